// jpeg_amd/csrc/huffdec.hip -- entropy DEcoding on gfx950.
//
// Replaces the reference's bit-serial loop
//     read_ecs (decoder.c:364-388) -> read_macroblock (:262-362) -> read_block (coeffs.c:196-254)
//       -> read_code (huffman.c:255-271: one next_bit() + one linear query_code() per bit)
//       -> read_extra_bits (huffman.c:294-312) -> decode_coeff (coeffs.c:33-48)
// which is >90 % of the reference's decode time, by a data-parallel scheme:
//
//   1. k_ecs_scan / k_ecs_chunk_scan / k_ecs_compact
//        byte un-stuffing (FF 00 -> FF, io.c:257-264) and RSTn segmentation (decoder.c:622-633) as
//        count -> scan -> scatter.  Output: one contiguous unstuffed bit stream per frame plus the
//        byte offset at which every restart segment starts.  The ECS ends at the first FF xx with
//        xx not in {00, D0..D7} (io.c:265-269).
//   2. k_huff_sync_intra
//        the stream is cut into sub-sequences of JH_SUBSEQ_BITS bits, one thread each.  Every thread
//        decodes its sub-sequence from the guess "a DC code of block 0 starts at my first bit" and
//        records its exit state (bit position, block-in-MCU, zig-zag index).  Huffman streams
//        self-synchronise: a wrong guess falls into step with the true parse after a few symbols.
//        Threads then keep decoding into the following sub-sequences and overwrite the stored exit
//        states until their own state equals the stored one (earlier-starting chains win; the first
//        sub-sequence of a frame, and everything after a restart boundary, is known to be right).
//   3. k_huff_sync_inter   the same across workgroup boundaries, iterated to a fixed point
//   4. k_huff_scan         exclusive (restart-segmented) prefix sums of blocks and DC differences
//   5. k_huff_write        every thread re-decodes its sub-sequence from its now-known entry state
//                          and scatters coefficients (absolute DC) into the int16 scan-order buffer.
//
// A restart boundary inside a sub-sequence resets the state (c = z = 0, predictors 0) whatever the
// state before was, so restart intervals and speculative decoding are ONE mechanism here: with
// markers every chain synchronises at the next marker at the latest; without them by self-synchronisation.
// The reference ignores Ri (quirk Q1); so does this code: segments are delimited by the markers found.
//
// Exactness: integer work; every symbol is decoded exactly as F.2.2 prescribes (canonical codes,
// first match = shortest match), so coefficients are bit-identical to the reference's.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"
#include "huffcore.hpp"

#define CHUNK_BYTES 4096              // raw bytes per workgroup in the un-stuffing kernels (256 x 16)

// ===================================================================================== 1. un-stuffing

// classification of one raw byte
//   bit0 kept (payload)   bit1 RSTn lead (FF of FF Dn)   bit2 terminator lead (FF xx, xx not 00/Dn)
__device__ __forceinline__ unsigned classify(unsigned prev, unsigned b, unsigned next)
{
	if (b == 0xffu) {
		if (next == 0x00u) {
			return 1u;                 // stuffed FF: data
		}
		if ((next & 0xf8u) == 0xd0u) {
			return 2u;                 // RSTn
		}
		return 4u;                     // any other marker (or fill byte): end of the entropy-coded data
	}
	if (prev == 0xffu && (b == 0x00u || (b & 0xf8u) == 0xd0u)) {
		return 0u;                     // stuffed zero / second byte of RSTn
	}
	return 1u;
}

struct RawWindow {
	unsigned bytes[18];               // [0] = byte before, [1..16] = own, [17] = byte after
};

// thread `tid` of chunk `chunk` owns absolute bytes [base, base + 16)
__device__ __forceinline__ void load_window(const uint8_t *raw, uint32_t lo, uint32_t hi, uint32_t base, RawWindow &w)
{
	// 16-byte aligned vector load where the whole window is inside the frame, byte loads at the edges
	if (base >= lo && base + 16 <= hi) {
		const uint4 v = *reinterpret_cast<const uint4 *>(raw + base);
		const unsigned ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			w.bytes[1 + i] = (ws[i >> 2] >> (8 * (i & 3))) & 0xffu;
		}
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t p = base + i;
			w.bytes[1 + i] = (p >= lo && p < hi) ? raw[p] : (p >= hi ? 0xd9u : 0u);
		}
	}
	w.bytes[0] = (base > lo && base - 1 < hi) ? raw[base - 1] : 0u;
	// a frame that ends in a lone FF: treat the missing byte as a marker code
	w.bytes[17] = (base + 16 < hi) ? raw[base + 16] : 0xd9u;
}

__device__ __forceinline__ unsigned block_reduce_add(unsigned v, unsigned *scratch)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		v += __shfl_down(v, o, 64);
	}
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (lane == 0) {
		scratch[wv] = v;
	}
	__syncthreads();
	unsigned r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
	__syncthreads();
	return r;
}

// exclusive prefix over the 256 threads of the block; total returned through *total
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *scratch, unsigned *total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		unsigned t = __shfl_up(inc, o, 64);
		if (lane >= o) {
			inc += t;
		}
	}
	if (lane == 63) {
		scratch[wv] = inc;
	}
	__syncthreads();
	unsigned base = 0;
	for (int i = 0; i < wv; ++i) {
		base += scratch[i];
	}
	*total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
	__syncthreads();
	return base + inc - v;
}

__global__ __launch_bounds__(256) void k_ecs_scan(JhDecodeWork w)
{
	__shared__ unsigned scratch[4];
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	if (blockIdx.x >= fr.nchunk) {
		return;
	}
	// chunks are aligned to absolute multiples of CHUNK_BYTES so that vector loads are aligned
	const uint32_t base = (fr.raw_begin / CHUNK_BYTES + blockIdx.x) * CHUNK_BYTES + threadIdx.x * 16;
	RawWindow win;
	load_window(w.raw, fr.raw_begin, fr.raw_end, base, win);
	unsigned kept = 0, rst = 0;
	uint32_t term = 0xffffffffu;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t p = base + i;
		if (p >= fr.raw_begin && p < fr.raw_end) {
			const unsigned cl = classify(win.bytes[i], win.bytes[i + 1], win.bytes[i + 2]);
			kept += cl & 1u;
			rst += (cl >> 1) & 1u;
			if ((cl & 4u) && term == 0xffffffffu) {
				term = p;
			}
		}
	}
	if (term != 0xffffffffu) {
		atomicMin(&w.state[f].term_pos, term);
	}
	const unsigned tk = block_reduce_add(kept, scratch);
	const unsigned tr = block_reduce_add(rst, scratch);
	if (threadIdx.x == 0) {
		w.chunk_kept[fr.chunk_off + blockIdx.x] = tk;
		w.chunk_rst[fr.chunk_off + blockIdx.x] = tr;
	}
}

// one workgroup per frame: exclusive scan of the per-chunk counts (in place)
__global__ __launch_bounds__(256) void k_ecs_chunk_scan(JhDecodeWork w)
{
	__shared__ unsigned scratch[4];
	const int f = blockIdx.x;
	const JhDecFrame fr = w.frames[f];
	unsigned carry_k = 0, carry_r = 0;
	for (uint32_t c0 = 0; c0 < fr.nchunk; c0 += 256) {
		const uint32_t c = c0 + threadIdx.x;
		const unsigned k = c < fr.nchunk ? w.chunk_kept[fr.chunk_off + c] : 0u;
		const unsigned r = c < fr.nchunk ? w.chunk_rst[fr.chunk_off + c] : 0u;
		unsigned tk, tr;
		const unsigned ek = block_excl_scan(k, scratch, &tk);
		const unsigned er = block_excl_scan(r, scratch, &tr);
		if (c < fr.nchunk) {
			w.chunk_kept[fr.chunk_off + c] = carry_k + ek;
			w.chunk_rst[fr.chunk_off + c] = carry_r + er;
		}
		carry_k += tk;
		carry_r += tr;
	}
	if (threadIdx.x == 0) {
		JhDecState &st = w.state[f];
		// totals for a stream without terminating marker; k_ecs_compact overrides them otherwise
		if (st.term_pos > fr.raw_end) {
			st.term_pos = fr.raw_end;
		}
		st.cs_len = carry_k;
		st.nseg = carry_r + 1;
		w.seg_start[fr.seg_off] = 0;
		if (carry_r + 1 < fr.seg_cap) {
			w.seg_start[fr.seg_off + carry_r + 1] = carry_k;
		}
	}
}

__global__ __launch_bounds__(256) void k_ecs_compact(JhDecodeWork w)
{
	__shared__ unsigned scratch[4];
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	if (blockIdx.x >= fr.nchunk) {
		return;
	}
	JhDecState &st = w.state[f];
	const uint32_t term = st.term_pos;
	const uint32_t base = (fr.raw_begin / CHUNK_BYTES + blockIdx.x) * CHUNK_BYTES + threadIdx.x * 16;
	RawWindow win;
	load_window(w.raw, fr.raw_begin, fr.raw_end, base, win);
	unsigned cls[16];
	unsigned kept = 0, rst = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t p = base + i;
		cls[i] = (p >= fr.raw_begin && p < fr.raw_end) ? classify(win.bytes[i], win.bytes[i + 1], win.bytes[i + 2]) : 0u;
		kept += cls[i] & 1u;
		rst += (cls[i] >> 1) & 1u;
	}
	unsigned tk, tr;
	unsigned ok = w.chunk_kept[fr.chunk_off + blockIdx.x] + block_excl_scan(kept, scratch, &tk);
	unsigned orr = w.chunk_rst[fr.chunk_off + blockIdx.x] + block_excl_scan(rst, scratch, &tr);
	uint8_t *cs = w.cs + fr.cs_off;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t p = base + i;
		if (p == term && p < fr.raw_end) {
			// the marker that ends the entropy-coded data: everything before it has been counted
			st.term_marker = win.bytes[i + 2];
			st.cs_len = ok;
			st.nseg = orr + 1;
			if (orr + 1 < fr.seg_cap) {
				w.seg_start[fr.seg_off + orr + 1] = ok;
			} else {
				atomicOr(&st.status, JH_ST_SEG_OVERFLOW);
			}
		}
		if (p < term) {
			if (cls[i] & 1u) {
				cs[ok] = (uint8_t)win.bytes[i + 1];
			}
			if (cls[i] & 2u) {
				// segment (orr + 1) starts at the payload byte that follows this marker
				if (orr + 1 < fr.seg_cap) {
					w.seg_start[fr.seg_off + orr + 1] = ok;
				} else {
					atomicOr(&st.status, JH_ST_SEG_OVERFLOW);
				}
			}
		}
		ok += cls[i] & 1u;
		orr += (cls[i] >> 1) & 1u;
	}
}

int jh_launch_ecs_unstuff(const JhDecodeWork &w, hipStream_t s, const JhTimerHook *th)
{
	dim3 grid(w.max_chunks, w.nframes);
	JH_T_BEGIN(th, "k_ecs_scan", s);
	hipLaunchKernelGGL(k_ecs_scan, grid, dim3(256), 0, s, w);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_ecs_chunk_scan", s);
	hipLaunchKernelGGL(k_ecs_chunk_scan, dim3(w.nframes), dim3(256), 0, s, w);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_ecs_compact", s);
	hipLaunchKernelGGL(k_ecs_compact, grid, dim3(256), 0, s, w);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ===================================================================================== 5. write pass

__global__ __launch_bounds__(256) void k_huff_write(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	uint32_t *sw = reinterpret_cast<uint32_t *>(smem + sizeof(JhHuffLut) * a.w.max_luts);
	__shared__ DecTables tb;
	__shared__ int dcs[JH_MAX_COMP][256];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t nsub = (total_bits + JH_SUBSEQ_BITS - 1) / JH_SUBSEQ_BITS;
	if (blockIdx.x * 256u >= nsub) {
		return;
	}
	const uint8_t *cs = w.cs + fr.cs_off;
#ifdef JH_LDS_STREAM
	LdsStream S;
	S.sw = sw;
	S.wbase = blockIdx.x * 256u * (JH_SUBSEQ_BITS / 32);
	stage_stream(sw, cs, S.wbase, min(256u, nsub - blockIdx.x * 256u) * (JH_SUBSEQ_BITS / 32) + 4);
#else
	(void)sw;
	const GlobalStream S{cs};
#endif
	load_tables(w, fr, luts, &tb, a.g);
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t >= nsub) {
		return;
	}
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	// entry record: relative to the chains workgroup that emitted it (huffsync.hip), plus that
	// workgroup's base.  blocks: plain sum (the reference's mblocks counter runs across restart markers,
	// decoder.c:381); DC predictors: back to 0 at every restart boundary (decoder.c:371-373)
	const JhSubStart s0 = w.start[fr.sub_off + t];
	const uint32_t cb = t < 256u ? 0u : (t - JH_CHAIN_MAX) / JH_SEQ_OWN;
	const JhWgAgg base = w.wg_base[fr.wg_off + cb];
	const bool cut = (s0.state >> 48) & 1;

	DecState st;
	st.pos = (uint32_t)(s0.state & 0xffffffffu);
	st.c = (int)((s0.state >> 32) & 0xff);
	st.z = (int)((s0.state >> 40) & 0xff);
	st.seg = s0.seg;
	st.reset = 0;
	st.nb = 0;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		st.dc[k] = cut ? s0.pred[k] : base.dc[k] + s0.pred[k];
	}
	const int blk0 = base.nb + s0.blk;
	if (st.c >= a.g.B || st.z > 63 || st.seg >= nseg || st.pos < t * JH_SUBSEQ_BITS) {
		return;                        // cannot happen for records produced by decode_span
	}
	decode_span<true>(st, (t + 1) * JH_SUBSEQ_BITS, S, total_bits, seg_start, nseg, luts, tb, a.g.B, &dcs[0][threadIdx.x], 256,
	                  w.coefs + (size_t)f * a.g.nblk * 64, blk0, a.g.nblk);
}

int jh_launch_huff_decode(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th)
{
	HuffArgs a;
	a.w = w;
	a.g = g;
	a.round = 0;
	// one launch configuration for the batch: LDS for the largest table set of any frame
	size_t lds = sizeof(JhHuffLut) * (size_t)w.max_luts;
#ifdef JH_LDS_STREAM
	lds += JH_STREAM_LDS_WORDS(256) * sizeof(uint32_t);
#endif
	dim3 gsub((w.max_subs + 255) / 256, w.nframes);
	if (jh_launch_huff_sync(w, g, s, th)) {
		return -1;
	}
	JH_T_BEGIN(th, "k_huff_write", s);
	hipLaunchKernelGGL(k_huff_write, gsub, dim3(256), lds, s, a);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
