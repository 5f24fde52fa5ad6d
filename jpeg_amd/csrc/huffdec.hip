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
//   2. huffsync.hip: k_huff_chains / k_huff_walk / k_huff_finish
//        the stream is cut into sub-sequences of 2^sub_shift bits (2048 as a rule), one thread each.  Every thread
//        decodes from the guess "a DC code of block 0 starts at my first bit"; Huffman streams
//        self-synchronise, so a wrong guess falls into step with the true parse after a few symbols.
//        The kernels find, for every sub-sequence, the state in which the TRUE parse enters it, and
//        the (restart-segmented) prefix sums of blocks and DC differences up to it.
//   3. k_huff_write   every thread re-decodes its sub-sequence from its now-known entry state,
//                     assembles whole 128-byte blocks (absolute DC) in LDS and the wave stores them.
//   4. k_coef_tail_clear   blocks the stream never reached are zero (common.c:129).
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

// ---- sixteen bytes at a time (SWAR) ---------------------------------------------------------------
// Threads whose 16 bytes, the byte before and the byte after all lie inside the frame -- all but two or
// three per frame -- classify them four at a time in 32-bit words: a flag in bit 7 of every byte.

__device__ __forceinline__ uint32_t swar_zero(uint32_t v)          // bit 7 of a byte set <=> the byte is 0x00 (exact)
{
	return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v | 0x7f7f7f7fu);
}

__device__ __forceinline__ uint32_t flags_to_bits(uint32_t f)       // bits 7, 15, 23, 31 -> bits 0 .. 3
{
	return (((f >> 7) * 0x00204081u) >> 21) & 0xfu;
}

struct Cls16 {
	uint32_t x[4];                    // the bytes (little endian words)
	uint32_t kept, rst, term;         // bit i = byte i: payload / FF of RSTn / FF of another marker
	uint32_t next;                    // the byte behind them
};

// fast: base > lo && base + 16 < hi (and base is a multiple of 16, so the words before and behind exist)
__device__ __forceinline__ void classify16_fast(const uint8_t *raw, uint32_t base, Cls16 &c)
{
	const uint4 v = *reinterpret_cast<const uint4 *>(raw + base);
	const uint32_t before = *reinterpret_cast<const uint32_t *>(raw + base - 4);
	const uint32_t behind = *reinterpret_cast<const uint32_t *>(raw + base + 16);
	c.x[0] = v.x;
	c.x[1] = v.y;
	c.x[2] = v.z;
	c.x[3] = v.w;
	c.next = behind & 0xffu;
	uint32_t ff[5], z0[5], dn[5];
	ff[0] = (before >> 24) == 0xffu ? 0x80000000u : 0u;             // only the byte before counts
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		ff[i + 1] = swar_zero(~c.x[i]);
		z0[i] = swar_zero(c.x[i]);
		dn[i] = swar_zero((c.x[i] & 0xf8f8f8f8u) ^ 0xd0d0d0d0u);
	}
	z0[4] = c.next == 0u ? 0x80u : 0u;                              // only the byte behind counts
	dn[4] = (c.next & 0xf8u) == 0xd0u ? 0x80u : 0u;
	uint32_t kept = 0, rst = 0, term = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const uint32_t f = ff[i + 1];
		const uint32_t n00 = __builtin_amdgcn_alignbit(z0[i + 1], z0[i], 8);     // the flags of every byte's successor
		const uint32_t ndn = __builtin_amdgcn_alignbit(dn[i + 1], dn[i], 8);
		const uint32_t pff = __builtin_amdgcn_alignbit(f, ff[i], 24);            // ... and of its predecessor
		const uint32_t k = (f & n00) | (~f & ~(pff & (z0[i] | dn[i])) & 0x80808080u);
		kept |= flags_to_bits(k) << (4 * i);
		rst |= flags_to_bits(f & ndn) << (4 * i);
		term |= flags_to_bits(f & ~(n00 | ndn)) << (4 * i);
	}
	c.kept = kept;
	c.rst = rst;
	c.term = term;
}

// any position: byte by byte, what lies outside [lo, hi) is neither kept nor a marker
__device__ __forceinline__ void classify16_edge(const uint8_t *raw, uint32_t lo, uint32_t hi, uint32_t base, Cls16 &c)
{
	RawWindow win;
	load_window(raw, lo, hi, base, win);
	c.x[0] = c.x[1] = c.x[2] = c.x[3] = 0;
	c.kept = c.rst = c.term = 0;
	c.next = win.bytes[17];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t p = base + i;
		c.x[i >> 2] |= win.bytes[i + 1] << (8 * (i & 3));
		if (p >= lo && p < hi) {
			const unsigned cl = classify(win.bytes[i], win.bytes[i + 1], win.bytes[i + 2]);
			c.kept |= (cl & 1u) << i;
			c.rst |= ((cl >> 1) & 1u) << i;
			c.term |= ((cl >> 2) & 1u) << i;
		}
	}
}

__device__ __forceinline__ void classify16(const uint8_t *raw, uint32_t lo, uint32_t hi, uint32_t base, Cls16 &c)
{
	if (base > lo && base + 16 < hi) {
		classify16_fast(raw, base, c);
	} else {
		classify16_edge(raw, lo, hi, base, c);
	}
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
	Cls16 c;
	classify16(w.raw, fr.raw_begin, fr.raw_end, base, c);
	if (c.term) {
		atomicMin(&w.state[f].term_pos, base + (uint32_t)__ffs(c.term) - 1u);
	}
	const unsigned both = block_reduce_add((unsigned)__popc(c.kept) | (unsigned)__popc(c.rst) << 16, scratch);
	if (threadIdx.x == 0) {
		w.chunk_kept[fr.chunk_off + blockIdx.x] = both & 0xffffu;      // <= 4096 and <= 2048: no carry between the halves
		w.chunk_rst[fr.chunk_off + blockIdx.x] = both >> 16;
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

// The payload bytes of a chunk, packed.  A thread squeezes the bytes it drops (the 00 behind an FF, RSTn, what
// lies behind the end of the data -- one or two in a sixteenth of the threads) out of its 16 bytes in
// registers, ORs what is left into an LDS image of the chunk's output laid out like the destination modulo 16,
// and the workgroup stores that image in whole aligned 16-byte pieces (the pieces it shares with the chunks
// before and behind byte by byte).
__global__ __launch_bounds__(256) void k_ecs_compact(JhDecodeWork w)
{
	__shared__ unsigned scratch[4];
	__shared__ __align__(16) uint32_t obuf[CHUNK_BYTES / 4 + 8];
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	if (blockIdx.x >= fr.nchunk) {
		return;
	}
	for (int i = threadIdx.x; i < CHUNK_BYTES / 4 + 8; i += 256) {
		obuf[i] = 0;
	}
	JhDecState &st = w.state[f];
	const uint32_t term = st.term_pos;
	const uint32_t base = (fr.raw_begin / CHUNK_BYTES + blockIdx.x) * CHUNK_BYTES + threadIdx.x * 16;
	Cls16 c;
	classify16(w.raw, fr.raw_begin, fr.raw_end, base, c);
	// nothing counts from the terminating marker on
	const uint32_t live = term <= base ? 0u : (term - base >= 16u ? 0xffffu : (1u << (term - base)) - 1u);
	const uint32_t kept = c.kept & live, rst = c.rst & live;
	const unsigned nk = (unsigned)__popc(kept);
	unsigned total;
	const unsigned excl = block_excl_scan(nk | (unsigned)__popc(rst) << 16, scratch, &total);   // ends with a barrier: obuf is zero
	const uint32_t ok0 = w.chunk_kept[fr.chunk_off + blockIdx.x];
	const uint32_t ok = ok0 + (excl & 0xffffu);
	const uint32_t orr = w.chunk_rst[fr.chunk_off + blockIdx.x] + (excl >> 16);

	if (term >= base && term < base + 16u && term < fr.raw_end) {
		// the marker that ends the entropy-coded data: everything before it has been counted
		const uint32_t i = term - base;
		st.term_marker = i == 15u ? c.next : (c.x[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu;
		st.cs_len = ok + nk;
		const uint32_t sg = orr + (uint32_t)__popc(rst) + 1u;
		st.nseg = sg;
		if (sg < fr.seg_cap) {
			w.seg_start[fr.seg_off + sg] = ok + nk;
		} else {
			atomicOr(&st.status, JH_ST_SEG_OVERFLOW);
		}
	}
	for (uint32_t m = rst, n = 1; m; m &= m - 1u, ++n) {
		// segment (orr + n) starts at the payload byte that follows this marker
		const uint32_t i = (uint32_t)__ffs(m) - 1u;
		if (orr + n < fr.seg_cap) {
			w.seg_start[fr.seg_off + orr + n] = ok + (uint32_t)__popc(kept & ((1u << i) - 1u));
		} else {
			atomicOr(&st.status, JH_ST_SEG_OVERFLOW);
		}
	}
	if (nk) {
		uint32_t x0 = c.x[0], x1 = c.x[1], x2 = c.x[2], x3 = c.x[3];
		for (uint32_t drop = ~kept & 0xffffu; drop;) {
			// squeeze out the highest dropped byte: everything above it moves down, a zero comes in at the top
			const uint32_t p = 31u - (uint32_t)__clz(drop);
			drop &= ~(1u << p);
			const uint32_t d = p >> 2, below = (1u << (8u * (p & 3u))) - 1u;
			const uint32_t s0 = __builtin_amdgcn_alignbit(x1, x0, 8), s1 = __builtin_amdgcn_alignbit(x2, x1, 8);
			const uint32_t s2 = __builtin_amdgcn_alignbit(x3, x2, 8), s3 = x3 >> 8;
			const uint32_t m0 = d == 0 ? below : 0xffffffffu;
			const uint32_t m1 = d < 1 ? 0u : (d == 1 ? below : 0xffffffffu);
			const uint32_t m2 = d < 2 ? 0u : (d == 2 ? below : 0xffffffffu);
			const uint32_t m3 = d < 3 ? 0u : below;
			x0 = (x0 & m0) | (s0 & ~m0);
			x1 = (x1 & m1) | (s1 & ~m1);
			x2 = (x2 & m2) | (s2 & ~m2);
			x3 = (x3 & m3) | (s3 & ~m3);
		}
		// into the image: byte q of it is byte (ok0 & ~15) + q of the frame's stream
		const uint32_t q = (ok0 & 15u) + (excl & 0xffffu);
		const uint32_t sh = 8u * (q & 3u);
		uint32_t *dst = obuf + (q >> 2);
		atomicOr(dst + 0, x0 << sh);
		atomicOr(dst + 1, (uint32_t)((((unsigned long long)x1 << 32 | x0) << sh) >> 32));
		atomicOr(dst + 2, (uint32_t)((((unsigned long long)x2 << 32 | x1) << sh) >> 32));
		atomicOr(dst + 3, (uint32_t)((((unsigned long long)x3 << 32 | x2) << sh) >> 32));
		atomicOr(dst + 4, (uint32_t)(((unsigned long long)x3 << sh) >> 32));
	}
	__syncthreads();
	const uint32_t nkept = total & 0xffffu;
	const uint32_t lead = ok0 & 15u;
	uint8_t *out = w.cs + fr.cs_off + (ok0 - lead);                // 16-byte aligned (cs_off is a multiple of 64)
	for (uint32_t g = threadIdx.x; g * 16u < lead + nkept; g += 256) {
		const uint32_t a = g * 16u < lead ? lead - g * 16u : 0u;    // this chunk's bytes in the piece: [a, b)
		const uint32_t b = min(16u, lead + nkept - g * 16u);
		if (a == 0 && b == 16u) {
			*reinterpret_cast<uint4 *>(out + g * 16u) = *reinterpret_cast<const uint4 *>(obuf + g * 4u);
		} else {
			for (uint32_t j = a; j < b; ++j) {
				out[g * 16u + j] = (uint8_t)(obuf[g * 4u + (j >> 2)] >> (8u * (j & 3u)));
			}
		}
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
//
// Every thread re-decodes its sub-sequence from its now-known entry state.  Coefficients go to HBM
// as WHOLE 128-byte blocks, each written exactly once: a thread collects the block it is decoding in
// its own 128-byte LDS buffer; when blocks complete, the wave flushes them together, 8 lanes per
// block with one 16-byte store each (a full cache line per block, no partial-line traffic, no
// separate zero-fill of the coefficient buffer).  A block that straddles a sub-sequence boundary
// belongs to the thread in whose sub-sequence it STARTS: that thread decodes on past its boundary
// until the block is complete, the next thread discards what it decodes before its first block start.

#define WR_THREADS 256                 // 128 B of LDS per thread + the tables once per workgroup: 3 workgroups = 12 waves
                                       // per CU.  448 threads (2 x 7 waves fill the CU's LDS) decode 3 % faster when the
                                       // decoder has the GPU to itself, but a step that encodes at the same time loses 5 %:
                                       // the encoder's workgroups then find no LDS beside them (bench `value` 80.3 against
                                       // 84.2 Gpix/s; 128 / 192 / 320 threads: 79.7 / 82.6 / 77.7)

// chains workgroup that emitted the entry record of sub-sequence t (huffsync.hip)
__device__ __forceinline__ uint32_t chain_wg_of(uint32_t t)
{
	return t < 256u ? 0u : (t - JH_CHAIN_MAX) / JH_SEQ_OWN;
}

struct WrEntry {
	uint32_t pos, seg;
	int c, z, blk;
	int dc[JH_MAX_COMP];
};

// entry record: relative to the chains workgroup that emitted it, plus that workgroup's base.
// blocks: plain sum (the reference's mblocks counter runs across restart markers, decoder.c:381);
// DC predictors: back to 0 at every restart boundary (decoder.c:371-373)
__device__ __forceinline__ WrEntry load_entry(const JhDecodeWork &w, const JhDecFrame &fr, uint32_t t)
{
	const JhSubStart s0 = w.start[fr.sub_off + t];
	const JhWgAgg base = w.wg_base[fr.wg_off + chain_wg_of(t)];
	const bool cut = (s0.state >> 48) & 1;
	WrEntry e;
	e.pos = (uint32_t)(s0.state & 0xffffffffu);
	e.c = (int)((s0.state >> 32) & 0xff);
	e.z = (int)((s0.state >> 40) & 0xff);
	e.seg = s0.seg;
	e.blk = base.nb + s0.blk;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		e.dc[k] = cut ? s0.pred[k] : base.dc[k] + s0.pred[k];
	}
	return e;
}

// block number in the scan's order -> block number in the coefficient buffer (JhRemap)
__device__ __forceinline__ int remap_block(const JhRemap &rm, int n)
{
	if (!rm.mode) {
		return n;
	}
	if (rm.mode == 2) {
		const int m = n / rm.nk;
		return m * rm.B + rm.kmap[n - m * rm.nk];
	}
	// raster position of the component -> (MCU, block of the MCU) of the frame
	const int by = n / rm.bxc, bx = n - by * rm.bxc;
	const int my = by / rm.V, mx = bx / rm.H;
	return (my * rm.m_x + mx) * rm.B + rm.kfirst + (by - my * rm.V) * rm.H + (bx - mx * rm.H);
}

// The wave writes the blocks its lanes completed in this iteration.  fin: this lane completed a block;
// fblk: its index in the frame, or -1 = discard (not this lane's block).  bufs = the workgroup's block
// buffers as 16-byte chunks, buffer of thread i = chunks [8 i, 8 i + 8), logical chunk j of thread i at
// physical position (j + i) & 7.  The flushed buffers are zeroed for the next block.
// pack: block numbers fit beside a lane number in one word (one exchange between the lanes instead of two)
__device__ __forceinline__ void flush_blocks(bool fin, int fblk, int4 *bufs, int4 *frame_coefs, bool pack)
{
	const unsigned long long m = __ballot(fin);
	if (m == 0) {
		return;
	}
	const int lane = threadIdx.x & 63;
	const int wave0 = (int)threadIdx.x - lane;
	const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
	const int n = __popcll(m);
	// a permutation of the lanes: the r-th finished lane hands its (block, lane) to lane r
	const int dst = fin ? below : n + (lane - below);
	int pblk, plane;
	if (pack) {
		pblk = __builtin_amdgcn_ds_permute(dst << 2, ((fblk + 1) << 6) | lane);
		plane = 0;
	} else {
		pblk = __builtin_amdgcn_ds_permute(dst << 2, fblk);
		plane = __builtin_amdgcn_ds_permute(dst << 2, lane);
	}
	for (int base = 0; base < n; base += 8) {
		const int g = base + (lane >> 3);
		int sblk, sthr;
		if (pack) {
			const int v = __builtin_amdgcn_ds_bpermute(g << 2, pblk);
			sblk = (v >> 6) - 1;
			sthr = wave0 + (v & 63);
		} else {
			sblk = __builtin_amdgcn_ds_bpermute(g << 2, pblk);
			sthr = wave0 + __builtin_amdgcn_ds_bpermute(g << 2, plane);
		}
		if (g < n) {
			// lane j of the group stores logical chunk j: the eight lanes write one whole 128-byte line
			// in ascending order (rotated orders reach L2 as partial sectors, which it fills from HBM first)
			const int j = lane & 7;
			int4 *src = bufs + sthr * 8 + ((j + sthr) & 7);
			const int4 v = *src;
			*src = make_int4(0, 0, 0, 0);
			if (sblk >= 0) {
				frame_coefs[(size_t)sblk * 8 + j] = v;
			}
		}
	}
}

// REMAP: the scan codes one component of several (JhRemap.mode 1); a kernel of its own, so that the usual
// case carries neither the test nor the registers.
//
// The loop is written for its instruction count (the kernel issues: 100 vector and 76 scalar instructions per
// symbol before this form): what happens once per sub-sequence or per restart segment -- opening the stream,
// the careful symbol near a segment's end, undecodable data -- sits behind tests the whole wave takes together,
// the symbol itself is straight-line code, and the DC predictors are registers (a returning LDS atomic per
// symbol made every iteration wait for the LDS once more).
// WIDE: sample precision 13 ... 16, where the DC predictions of a valid stream can add up past the 16 bits the
// coefficients are stored in: such a frame is reported (JH_ST_COEF_RANGE), not stored wrapped.
template <bool REMAP, bool WIDE>
__global__ __launch_bounds__(WR_THREADS) void k_huff_write(const HuffArgs a)
{
	extern __shared__ __align__(16) unsigned char smem[];
	unsigned char *luts = smem;
	int4 *bufs = reinterpret_cast<int4 *>(smem + a.w.lut_bytes);
	__shared__ DecTables tb;
	__shared__ int dcs_all[JH_MAX_COMP][WR_THREADS];

	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhDecFrame fr = w.frames[f];
	const JhDecState fs = w.state[f];
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t sub_shift = w.sub_shift;
	const uint32_t nsub = (total_bits + (1u << sub_shift) - 1u) >> sub_shift;
	const uint32_t t0 = blockIdx.x * (uint32_t)WR_THREADS;
	if (t0 >= nsub) {
		return;
	}
	const int tid = threadIdx.x;
	const uint8_t *cs = w.cs + fr.cs_off;
	const RegStream S{cs};
	const GlobalStream G{cs};
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		bufs[tid * 8 + j] = make_int4(0, 0, 0, 0);
	}
	load_tables(w, fr, luts, &tb, a.g, WR_THREADS * 4);  // ends with a barrier

	const uint32_t t = t0 + tid;
	const uint32_t *seg_start = w.seg_start + fr.seg_off;
	const uint32_t nseg = min(fs.nseg, fr.seg_cap - 1);
	const int B = a.g.B, nblk = a.g.nblk;
	const JhRemap rm = w.rm;
	const bool pack = (REMAP ? rm.nblk_full : nblk) < (1 << 25);
	int4 *frame_coefs = reinterpret_cast<int4 *>(w.coefs + (size_t)f * (REMAP ? rm.nblk_full : nblk) * 64);
	int16_t *mybuf = reinterpret_cast<int16_t *>(bufs + tid * 8);
	const int rot = (tid & 7) << 3;                  // coefficient k of the block in progress sits at mybuf[(k + rot) & 63]

	// ---- entry of this thread ------------------------------------------------------------------------
	bool act = t < nsub;
	uint32_t pos = 0, seg = 0, seg_end = 0;
	int c = 0, z = 0, blk = 0;
	int *dcs = &dcs_all[0][tid];                     // DC predictors by component (decoder.c:351-355)
	if (act) {
		const WrEntry e = load_entry(w, fr, t);
		pos = e.pos;
		seg = e.seg;
		c = e.c;
		z = e.z;
		blk = e.blk;
#pragma unroll
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			DCS_AT(dcs, k, WR_THREADS) = e.dc[k];
		}
		// cannot fail for records produced by decode_span
		act = c < B && z <= 63 && seg < nseg && pos >= (t << sub_shift) && pos < total_bits;
		if (act) {
			seg_end = min(seg_start[seg + 1] * 8u, total_bits);
		}
	}
	const uint32_t stop = ((t + 1) << sub_shift);
	// the thread's flags live in ONE vector register: kept as wave masks they cost a dozen scalar instructions per
	// symbol to merge at the ends of the divergent regions, and the scalar unit is shared by the CU's four SIMDs
	enum { F_ACT = 1, F_OWN = 2, F_PRIMED = 4 };
	unsigned fl = (act ? F_ACT : 0u) | (z == 0 ? F_OWN : 0u);   // a thread that enters mid-block does not own that block
	// fast-path registers (see decode_span)
	int cn = 0;
	uint4 ent_n = make_uint4(0, 0, 0, 0);
	uint32_t wi = 0, w0 = 0, w1 = 0, w2 = 0, win = 0, tac = 0, tab = 0, comp = 0;
	int lastold = 0;                                 // DC predictor that went into the block in progress
	int wide = 0;
	RegStream::Cursor cur;
	cur.b = cur.c = cur.d = 0;
	cur.nq = 4;
	cur.nxt = 0;
	cur.ld.x = cur.ld.y = cur.ld.z = cur.ld.w = 0;
	unsigned e = 0;

	for (;;) {
		const bool go = (fl & F_ACT) && (pos < stop || z != 0);
		if (__builtin_amdgcn_ballot_w64(go) == 0) {
			break;
		}
		bool fin = false;
		int fblk = -1;
		// symbols that start at least 31 bits before the end of the restart segment cannot run past it
		bool fast = go && pos < (seg_end > 31u ? seg_end - 31u : 0u);
		if (__builtin_amdgcn_ballot_w64(fast && !(fl & F_PRIMED)) != 0) {
			if (fast && !(fl & F_PRIMED)) {
				const uint4 ent = tb.ent[c];
				cn = (int)ent.w;
				ent_n = tb.ent[cn];
				tac = ent.y;
				comp = ent.z;
				wi = pos >> 5;
				S.open(cur, wi, w0, w1, w2);
				win = bit_window(w0, w1, pos);
				tab = z == 0 ? ent.x : tac;
				e = lut_lookup(luts + tab, win);
				fl |= F_PRIMED;
			}
		}
		if (fast && (e & 0x8000u)) {
			e = long_code(luts + tab, e, win, z == 0);
			fast = e != 0;                     // no code matches: the careful path ends the segment
		}
		if (fast) {
			const bool isdc = z == 0;
			const int total1 = (int)((e >> 10) & 31u), run = (int)((e >> 4) & 63u), cat = (int)(e & 15u);
			// The symbol behind it as well, if the table has resolved it (JhHuffLut) and the first one leaves the block
			// open: two coefficients per trip.  (A DC symbol is never one of two; a thread runs on to the end of the
			// block it is in, so a pair never has to stop at the sub-sequence's end.)
			const unsigned pr = e >> 16;
			const int k1 = z + run;                                 // zero run (coeffs.c:246)
			const bool two = pr != 0 && !isdc && k1 < 63;
			const int total = two ? (int)JH_PAIR_TOTAL(pr) : total1;
			const uint32_t npos = pos + (uint32_t)total;
			const int k = two ? z + (int)JH_PAIR_ADV(pr) : k1;
			const bool done = k >= 63;
			// the next symbol: its table, its window, its look-up
			const uint32_t tabn = done ? ent_n.x : tac;
			const uint32_t nwi = npos >> 5;
			const bool cross = nwi != wi;
			w0 = cross ? w1 : w0;
			w1 = cross ? w2 : w1;
			wi = nwi;
			const uint32_t nwin = bit_window(w0, w1, npos);
			const unsigned en = lut_lookup(luts + tabn, nwin);
			w2 = S.next(cur, cross, w2);
			// the first symbol's value; DC un-prediction (decoder.c:351-355)
			const unsigned extra = __builtin_amdgcn_ubfe(win, 32u - (unsigned)total1, (unsigned)cat);
			int val = extend_sign(extra, cat);
			const int old = atomicAdd(reinterpret_cast<int *>(reinterpret_cast<char *>(dcs) + comp), isdc ? val : 0);
			val += isdc ? old : 0;
			lastold = isdc ? old : lastold;
			if (WIDE) {
				wide |= isdc ? (val + 32768) : 0;                   // bit 16 and up, or negative: not an int16
			}
			// coefficient k1 of the block in progress; EOB and ZRL carry a zero, a run past the end (k1 > 63) stores one
			mybuf[(min(k1, 63) + rot) & 63] = (int16_t)(k1 < 64 ? val : 0);
			if (two) {
				// the second symbol's: its magnitude bits are the last cat2 of the pair's bits
				const int cat2 = (int)JH_PAIR_CAT2(pr);
				const unsigned extra2 = __builtin_amdgcn_ubfe(win, 32u - (unsigned)total, (unsigned)cat2);
				const int val2 = extend_sign(extra2, cat2);
				mybuf[(min(k, 63) + rot) & 63] = (int16_t)(k < 64 ? val2 : 0);
			}
			z = done ? 0 : k + 1;
			fin = done;
			fblk = (done && (fl & F_OWN) && (unsigned)blk < (unsigned)nblk) ? blk : -1;
			if (REMAP && fblk >= 0) {
				fblk = remap_block(rm, fblk);
			}
			fl |= done ? (unsigned)F_OWN : 0u;
			blk += done ? 1 : 0;
			c = done ? cn : c;
			tac = done ? ent_n.y : tac;
			comp = done ? ent_n.z : comp;
			cn = done ? (int)ent_n.w : cn;
			ent_n = tb.ent[cn];
			pos = npos;
			win = nwin;
			tab = tabn;
			e = en;
		}
		if (__builtin_amdgcn_ballot_w64(go && !fast) != 0) {
			if (go && !fast) {
				// ---- careful: one symbol with every test of read_block(), stream read from global memory
				fl &= ~(unsigned)F_PRIMED;
				const bool isdc = z == 0;
				const uint32_t info = tb.info[c];
				const uint32_t kc = info >> 24;
				const unsigned char *tb1 = luts + (((info >> (isdc ? 0 : 12)) & 0xfffu) << 4);
				const uint32_t wj = pos >> 5;
				const uint32_t wn = bit_window(G.word(wj), G.word(wj + 1), pos);
				unsigned e1 = reinterpret_cast<const uint16_t *>(tb1)[wn >> (32 - JH_LUT_BITS)];
				if (e1 & 0x8000u) {
					e1 = long_code(tb1, e1, wn, isdc);
				}
				const int total = (int)(e1 >> 10), run = (int)((e1 >> 4) & 63u), cat = (int)(e1 & 15u);
				const uint32_t npos = pos + (uint32_t)total;
				if (e1 == 0 || npos > seg_end) {
					// End of this restart segment (decoder.c:376-381): the MCU in progress is not counted, the next
					// segment starts at its first block with DC predictors 0 (decoder.c:369-373).  What the reference
					// has stored of that MCU stays until the next segment overwrites it: the blocks it completed
					// (flushed already), and the block it was reading when the data ran out, once its DC symbol was
					// in -- with the DC DIFFERENCE, since the prediction is added after read_block() has succeeded
					// (coeffs.c:222, decoder.c:347-353), and the AC coefficients read so far.
					// The next segment's blocks go to the same places, stored by other threads: k_huff_fixup settles
					// the order afterwards.  Recorded by the thread in whose sub-sequence the segment ends.
					const bool partial = z != 0;
					// (pos >= total_bits: the thread is past its sub-sequence and no other starts behind the data)
					if ((pos < stop || pos >= total_bits) && (c != 0 || partial)) {
						if (seg + 1 < nseg) {
							w.drop[fr.seg_off + seg + 1] = ((unsigned long long)(uint32_t)(blk - c) << 8) | (unsigned)(c + (partial ? 1 : 0));
							atomicAdd(&w.state[f].ndrop, 1u);
						}
						// what is left of the MCU stays unless a later segment stores over it (none may: the last
						// segment, or segments that do not get as far): k_coef_tail_clear keeps off
						atomicMax(&w.state[f].ntail, (uint32_t)max(blk + (partial ? 1 : 0), 0));
					}
					if (partial) {
						mybuf[rot] -= (int16_t)lastold;             // coefficient 0
						fin = true;
						fblk = ((fl & F_OWN) && blk >= 0 && blk < nblk) ? blk : -1;
						if (REMAP && fblk >= 0) {
							fblk = remap_block(rm, fblk);
						}
					}
					blk -= c;
					c = 0;
					z = 0;
					fl |= F_OWN;
#pragma unroll
					for (int q = 0; q < JH_MAX_COMP; ++q) {
						DCS_AT(dcs, q, WR_THREADS) = 0;
					}
					seg++;
					if (seg >= nseg) {
						fl &= ~(unsigned)F_ACT;
					} else {
						pos = seg_end;
						seg_end = min(seg_start[seg + 1] * 8u, total_bits);
					}
				} else {
					const unsigned extra = __builtin_amdgcn_ubfe(wn, 32u - (unsigned)total, (unsigned)cat);
					int val = extend_sign(extra, cat);
					if (isdc) {
						lastold = atomicAdd(dcs + kc * WR_THREADS, val);
						val += lastold;
						if (WIDE) {
							wide |= val + 32768;
						}
					}
					const int k = z + run;
					const bool done = k >= 63;
					mybuf[(min(k, 63) + rot) & 63] = (int16_t)(k < 64 ? val : 0);
					pos = npos;
					z = done ? 0 : k + 1;
					if (done) {
						c = (c + 1 == B) ? 0 : c + 1;
						fin = true;
						fblk = ((fl & F_OWN) && blk >= 0 && blk < nblk) ? blk : -1;
						if (REMAP && fblk >= 0) {
							fblk = remap_block(rm, fblk);
						}
						fl |= F_OWN;
						blk++;
					}
				}
			}
		}
		flush_blocks(fin, fblk, bufs, frame_coefs, pack);
	}
	// The data of the LAST segment may end exactly behind a block that is not the last of its MCU, and behind this
	// thread's sub-sequence: no thread starts there (there is no bit left), so nobody meets the end of the segment.
	// The blocks of that unfinished MCU stay (decoder.c:347-355): say so before k_coef_tail_clear asks.
	if ((fl & F_ACT) && z == 0 && c != 0 && pos >= total_bits) {
		atomicMax(&w.state[f].ntail, (uint32_t)max(blk, 0));
	}
	if (WIDE && ((unsigned)wide >> 16) != 0) {
		atomicOr(&w.state[f].status, (uint32_t)JH_ST_COEF_RANGE);
	}
}

// blocks the stream never reached (truncated data, frames that failed to parse) are zero, as in
// alloc_buffers() (common.c:129); normally there is none
__global__ __launch_bounds__(256) void k_coef_tail_clear(const HuffArgs a)
{
	const JhDecodeWork &w = a.w;
	const int f = blockIdx.y;
	const size_t nblk = (size_t)a.g.nblk;
	const size_t first = min((size_t)max(w.state[f].nblocks, w.state[f].ntail), nblk);
	int4 *p = reinterpret_cast<int4 *>(w.coefs + (size_t)f * nblk * 64);
	for (size_t i = first * 8 + blockIdx.x * 256u + threadIdx.x; i < nblk * 8; i += (size_t)gridDim.x * 256u) {
		p[i] = make_int4(0, 0, 0, 0);
	}
}

// A restart segment that ends inside an MCU (truncated data, or an interval that is no multiple of the
// reference's block group in a single-component scan): the reference does not count that MCU
// (decoder.c:376-381), so the next segment's first blocks land where its blocks -- the completed ones and
// the one the data ran out in -- were stored.  In k_huff_write both are stored by unrelated threads; here the
// later ones are decoded once more, in segment order, after everything else.  One wave per frame, lane 0 decodes: such streams are rare,
// a frame without them costs one load.
// The first `count` blocks of the segment at [pos, seg_end) go to blocks first, first + 1, ...; returns how many
// of them were complete inside the segment.
__device__ int fixup_segment(const HuffArgs &a, const JhDecFrame *fr, int16_t *fc, const GlobalStream &G,
                             const unsigned char *luts, uint32_t pos, const uint32_t seg_end, int first, int count)
{
	const JhRemap rm = a.w.rm;
	int pred[JH_MAX_COMP];
#pragma unroll
	for (int q = 0; q < JH_MAX_COMP; ++q) {
		pred[q] = 0;
	}
	for (int c = 0; c < count; ++c) {
		const int n = first + c;
		const unsigned char *tdc = luts + (size_t)fr->blk_dc[c] * sizeof(JhHuffLut);
		const unsigned char *tac = luts + (size_t)fr->blk_ac[c] * sizeof(JhHuffLut);
		const int comp = a.g.blk_comp[c];
		int16_t *dst = n < a.g.nblk ? fc + (size_t)remap_block(rm, n) * 64 : nullptr;
		int dcv = 0;
		int z = 0;
		// read_block() (coeffs.c:196-254): nothing is touched before the DC symbol is in; then the block is
		// reset and holds the DC difference; the AC coefficients arrive one by one; the prediction is added
		// once the block is complete (decoder.c:351-355).  Wherever the segment ends, what is there stays.
		for (;;) {
			const bool isdc = z == 0;
			const unsigned char *tab = isdc ? tdc : tac;
			const uint32_t wj = pos >> 5;
			const uint32_t wn = bit_window(G.word(wj), G.word(wj + 1), pos);
			unsigned e = reinterpret_cast<const uint16_t *>(tab)[wn >> (32 - JH_LUT_BITS)];
			if (e & 0x8000u) {
				e = long_code(tab, e, wn, isdc);
			}
			const int total = (int)(e >> 10), run = (int)((e >> 4) & 63u), cat = (int)(e & 15u);
			if (e == 0 || pos + (uint32_t)total > seg_end) {
				return c;
			}
			const int val = extend_sign(__builtin_amdgcn_ubfe(wn, 32u - (unsigned)total, (unsigned)cat), cat);
			const int k = z + run;
			if (isdc) {
				dcv = val;
				if (dst) {
					for (int j = 0; j < 8; ++j) {
						reinterpret_cast<int4 *>(dst)[j] = make_int4(0, 0, 0, 0);
					}
					dst[0] = (int16_t)val;
				}
			} else if (dst && k < 64 && cat != 0) {
				dst[k] = (int16_t)val;
			}
			pos += (uint32_t)total;
			z = k + 1;
			if (k >= 63) {
				break;
			}
		}
#pragma unroll
		for (int q = 0; q < JH_MAX_COMP; ++q) {
			dcv += q == comp ? pred[q] : 0;
		}
		if (dst) {
			dst[0] = (int16_t)dcv;
		}
		if (dcv < -32768 || dcv > 32767) {
			atomicOr(&a.w.state[blockIdx.x].status, (uint32_t)JH_ST_COEF_RANGE);   // P > 12 only (see k_huff_write)
		}
#pragma unroll
		for (int q = 0; q < JH_MAX_COMP; ++q) {
			pred[q] = q == comp ? dcv : pred[q];
		}
	}
	return count;
}

__global__ __launch_bounds__(64) void k_huff_fixup(const HuffArgs a)
{
	const JhDecodeWork &w = a.w;
	const int f = blockIdx.x;
	const JhDecState fs = w.state[f];
	if (fs.ndrop == 0) {
		return;
	}
	const JhDecFrame *fr = w.frames + f;
	const uint32_t nseg = min(fs.nseg, fr->seg_cap - 1);
	const uint32_t total_bits = fs.cs_len * 8u;
	const uint32_t *seg_start = w.seg_start + fr->seg_off;
	const unsigned long long *drop = w.drop + fr->seg_off;
	const unsigned char *luts = reinterpret_cast<const unsigned char *>(w.luts + fr->lut_off);
	const GlobalStream G{w.cs + fr->cs_off};
	int16_t *fc = w.coefs + (size_t)f * (w.rm.mode ? w.rm.nblk_full : a.g.nblk) * 64;
	const int lane = threadIdx.x;
	uint32_t handled = 0;                                  // segments below this index are settled
	for (uint32_t s0 = 0; s0 < nseg; s0 += 64) {
		const uint32_t s = s0 + lane;
		const unsigned long long ev = s < nseg ? drop[s] : 0ull;
		unsigned long long m = __ballot(ev != 0);
		while (m) {
			const int src = __builtin_ctzll(m);
			m &= m - 1;
			const uint32_t lo = __shfl((uint32_t)ev, src, 64), hi = __shfl((uint32_t)(ev >> 32), src, 64);
			if (lane == 0) {
				const unsigned long long e = ((unsigned long long)hi << 32) | lo;
				const int first = (int)(e >> 8);
				// Segment sg starts over the MCU its predecessor left unfinished.  While a segment does not
				// complete that MCU either, the one behind it starts there again (context->mblocks stands still,
				// decoder.c:376-381): the whole run is redone here in stream order, each segment's first MCU
				// (or what there is of it) over what the earlier ones left.
				for (uint32_t sg = s0 + (uint32_t)src; sg >= handled && sg < nseg; ++sg) {
					const int done = fixup_segment(a, fr, fc, G, luts, seg_start[sg] * 8u, min(seg_start[sg + 1] * 8u, total_bits),
					                               first, a.g.B);
					handled = sg + 1;
					if (done == a.g.B) {
						break;
					}
				}
			}
			handled = __shfl(handled, 0, 64);
		}
	}
}

int jh_launch_huff_decode(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th)
{
	HuffArgs a;
	a.w = w;
	a.g = g;
	a.round = 0;
	if (jh_launch_huff_sync(w, g, s, th)) {
		return -1;
	}
	// one launch configuration for the batch: LDS for the largest table set of any frame
	const size_t lds = (size_t)w.lut_bytes + (size_t)WR_THREADS * 128;
	const bool wide = g.P > 12;
	void (*wr)(const HuffArgs) = w.rm.mode ? (wide ? k_huff_write<true, true> : k_huff_write<true, false>)
	                                       : (wide ? k_huff_write<false, true> : k_huff_write<false, false>);
	if (hipFuncSetAttribute(reinterpret_cast<const void *>(wr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
		return -1;
	}
	dim3 gsub((w.max_subs + WR_THREADS - 1) / WR_THREADS, w.nframes);
	JH_T_BEGIN(th, "k_huff_write", s);
	hipLaunchKernelGGL(wr, gsub, dim3(WR_THREADS), lds, s, a);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_huff_fixup", s);
	hipLaunchKernelGGL(k_huff_fixup, dim3(w.nframes), dim3(64), 0, s, a);
	JH_T_END(th, s);
	if (w.rm.mode == 0) {                              // scans of single components: the caller cleared the buffer
		JH_T_BEGIN(th, "k_coef_tail_clear", s);
		hipLaunchKernelGGL(k_coef_tail_clear, dim3(64, w.nframes), dim3(256), 0, s, a);
		JH_T_END(th, s);
	}
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
