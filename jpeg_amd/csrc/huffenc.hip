// jpeg_amd/csrc/huffenc.hip -- entropy ENcoding on gfx950.
//
// Replaces the reference's serial loops
//     write_ecs_dry (encoder.c:525-558) -> write_macroblock_dry (:466-518) -> write_block_dry (coeffs.c:312-363)
//     write_ecs     (encoder.c:560-587) -> write_macroblock (:412-464)     -> write_block (coeffs.c:256-309)
//       -> write_code (huffman.c:274-292) / write_extra_bits (:314-325) -> put_bit (io.c:43-63, one
//          fwrite per byte) -> write_ecs_byte (io.c:277-290) -> flush_bits (io.c:65-87)
// by data-parallel stages over the int16 scan-order coefficient buffer:
//
//   k_enc_histogram  symbol statistics per table (LDS histograms, one global atomic per used bin)
//   k_enc_dcbits     default tables: the pixel kernel has already counted the bits of every block's AC symbols
//                    (while the block was in LDS); this adds the DC symbol (its predictor lives in
//                    another block), the workgroup sums and the two conditions the reference assert()s / fails on
//   k_enc_bitlen     optimised tables or coefficients from the caller: the same from the coefficients
//                    (Figure F.2 walk with the table's code sizes), nonzero maps and DC values written on the way
//   k_enc_wg_scan    exclusive scan of the workgroup sums = bit offset of every 256-block group
//   k_enc_pack       every thread re-walks its block and ORs its code words into an LDS image of the
//                    workgroup's bit range (MSB first); the image is stored coalesced, the two shared
//                    boundary words merged with atomicOr
//   k_enc_ffcount / k_enc_ff_scan / k_enc_stuff
//                    0xFF -> 0xFF 0x00 byte stuffing as count -> scan -> scatter, final 1-padding,
//                    EOI appended
//
// The whole image is ONE entropy-coded segment, as in the reference (no restart markers: quirk Q2).
// Integer work: the produced bytes are identical to write_ecs()'s.
//
// EXTENSION (restart_interval > 0): DRI + RSTn.  Every interval starts on a byte
// boundary, so its padding depends only on its own bit count: k_enc_ivl_ends / k_enc_ivl_scan turn the
// per-block bit counts into per-interval paddings and their prefix sums; k_enc_pack shifts every block by
// the padding before its interval and appends the 1-bits to the interval's last block; the stuffing
// kernels insert FF D0+(k mod 8) at the interval ends (markers are not stuffed).  Judge: the reference
// decoder decodes these streams to the same samples (tests).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "common.hpp"
#include "kernels.hpp"

#define WIN_WORDS 2048                // LDS image of one workgroup's bit range: 8 KB = 256 bits/block per pass
#define ST_STRIDE 33                  // words per staged block (32 + 1: lanes reading one column hit 64 banks)
#define FF_CHUNK 4096

struct EncArgs {
	JhEncodeWork w;
	JhGeom g;
};

// ---------------------------------------------------------------------------------------- block walk

struct BlockRegs {
	int w[32];                        // 64 int16 coefficients, zig-zag order, two per dword
};

__device__ __forceinline__ void load_block(const int16_t *src, BlockRegs &b)
{
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int4 v = reinterpret_cast<const int4 *>(src)[j];
		b.w[4 * j] = v.x;
		b.w[4 * j + 1] = v.y;
		b.w[4 * j + 2] = v.z;
		b.w[4 * j + 3] = v.w;
	}
}

__device__ __forceinline__ int coef_at(const BlockRegs &b, int i)   // i is a compile-time constant after unrolling
{
	return (i & 1) ? (b.w[i >> 1] >> 16) : (int)(short)(b.w[i >> 1] & 0xffff);
}

// encode_cat (coeffs.c:56-74): number of bits of |c|
__device__ __forceinline__ int category(int c)
{
	const int m = c < 0 ? -c : c;
	return 32 - __clz(m);             // __clz(0) = 32
}

// leading bits of v that equal its sign bit, the sign bit included (v is neither 0 nor -1)
__device__ __forceinline__ int sign_bits(int v)
{
	int n;
	asm("v_ffbh_i32 %0, %1" : "=v"(n) : "v"(v));
	return n;
}

// encode_extra (coeffs.c:76-83)
__device__ __forceinline__ unsigned extra_bits(int c, int cat)
{
	return (unsigned)(c < 0 ? c - 1 : c) & ((1u << cat) - 1u);
}

// DC prediction as write_macroblock does it (encoder.c:443-458): the previous block of the SAME
// component in scan order; 0 at the start of the scan (and of a restart interval, extension).
__device__ __forceinline__ int dc_predictor(const int16_t *frame_coefs, int blk, const JhGeom &g, int ri)
{
	const int mcu = blk / g.B, k = blk - mcu * g.B;
	const int comp = g.blk_comp[k];
	if (k > g.comp_first[comp]) {
		return frame_coefs[(size_t)(blk - 1) * 64];
	}
	if (mcu == 0 || (ri > 0 && mcu % ri == 0)) {
		return 0;
	}
	const int last = g.comp_first[comp] + g.H[comp] * g.V[comp] - 1;
	return frame_coefs[((size_t)(mcu - 1) * g.B + last) * 64];
}

// the same from the per-block records (no 128-byte line touched for two bytes)
__device__ __forceinline__ int dc_predictor_info(const uint32_t *acdc, int blk, const JhGeom &g, int ri)
{
	const int mcu = blk / g.B, k = blk - mcu * g.B;
	const int comp = g.blk_comp[k];
	if (k > g.comp_first[comp]) {
		return (int)acdc[blk - 1] >> 16;
	}
	if (mcu == 0 || (ri > 0 && mcu % ri == 0)) {
		return 0;
	}
	const int last = g.comp_first[comp] + g.H[comp] * g.V[comp] - 1;
	return (int)acdc[(size_t)(mcu - 1) * g.B + last] >> 16;
}

// bit 0: the low int16 of d is nonzero, bit 1: the high one
__device__ __forceinline__ unsigned nz_pair(int d)
{
	typedef unsigned short us2 __attribute__((ext_vector_type(2)));
	const us2 one = {1, 1};
	const unsigned x = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, d), one));
	return (x | (x >> 15)) & 3u;
}

__device__ __forceinline__ unsigned long long block_nzmap(const BlockRegs &b)
{
	unsigned lo = 0, hi = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		lo |= nz_pair(b.w[i]) << (2 * i);
		hi |= nz_pair(b.w[16 + i]) << (2 * i);
	}
	return ((unsigned long long)hi << 32) | lo;
}

// Figure F.2 (coeffs.c:276-306): calls sym(rs, cat, extra) for every AC symbol of the block.
template <typename F>
__device__ __forceinline__ void walk_ac(const BlockRegs &b, F &&sym)
{
	int r = 0;
#pragma unroll
	for (int i = 1; i < 64; ++i) {
		const int c = coef_at(b, i);
		if (c == 0) {
			if (i == 63) {
				sym(0, 0, 0u);                             // EOB
			} else {
				r++;
			}
		} else {
			while (r > 15) {
				sym(0xf0, 0, 0u);                          // ZRL
				r -= 16;
			}
			const int cat = category(c);
			sym((r << 4) | (cat & 15), cat, extra_bits(c, cat));
			r = 0;
		}
	}
}

// ---------------------------------------------------------------------------------------- histogram

__global__ __launch_bounds__(256) void k_enc_histogram(const EncArgs a)
{
	__shared__ unsigned hist[4][257];
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.y;
	for (int i = threadIdx.x; i < 4 * 257; i += 256) {
		(&hist[0][0])[i] = 0;
	}
	__syncthreads();
	const int blk = blockIdx.x * 256 + threadIdx.x;
	const int16_t *fc = w.coefs + (size_t)f * g.nblk * 64;
	if (blk < g.nblk) {
		BlockRegs b;
		load_block(fc + (size_t)blk * 64, b);
		const int comp = g.blk_comp[blk % g.B];
		const int diff = coef_at(b, 0) - dc_predictor(fc, blk, g, w.restart_interval);
		unsigned *hd = hist[0 * 2 + (g.Td[comp] & 1)];
		unsigned *ha = hist[1 * 2 + (g.Ta[comp] & 1)];
		atomicAdd(&hd[category(diff) & 15], 1u);           // coeffs.c:327
		walk_ac(b, [&](int rs, int, unsigned) { atomicAdd(&ha[rs], 1u); });
	}
	__syncthreads();
	unsigned *gh = w.hist + (size_t)f * 4 * 257;
	for (int i = threadIdx.x; i < 4 * 257; i += 256) {
		const unsigned v = (&hist[0][0])[i];
		if (v) {
			atomicAdd(&gh[i], v);
		}
	}
}

int jh_launch_enc_histogram(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th)
{
	EncArgs a;
	a.w = w;
	a.g = g;
	hipMemsetAsync(w.hist, 0, (size_t)w.nframes * 4 * 257 * sizeof(uint32_t), s);
	JH_T_BEGIN(th, "k_enc_histogram", s);
	hipLaunchKernelGGL(k_enc_histogram, dim3(w.nwg, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------- table adaptation

// adapt_huffman_table (huffman.c:508-537) on the device, one wave per (frame, table): no trip to the host
// between the statistics and the bit packing.
//   Figure K.1 (huffman.c:366-411): the two least frequent entries are merged until one is left; among
//     equals the LARGEST index is taken (huffman.c:338).  The reference walks two linked lists to add one
//     to the code size of every member of both trees; here every symbol carries the index of its tree's
//     head, and the lanes update the members of both trees side by side.
//   Figures K.2-K.4 (huffman.c:413-506): sizes counted, limited to 16 bits, the reserved all-ones code
//     point removed, symbols listed by (size, value).
//   Then Annex C codes in the first-match sense of value_to_vlc (huffman.c:228-253).
__global__ __launch_bounds__(64) void k_enc_adapt(const EncArgs a)
{
	__shared__ uint32_t freq[320];
	__shared__ uint16_t head[320];
	__shared__ uint8_t csize[320];
	__shared__ int bits[34];
	__shared__ uint8_t outv[256];
	__shared__ int first_idx[18];
	__shared__ uint32_t first_code[18];
	const JhEncodeWork &w = a.w;
	const int t = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
	if ((t & 1) >= w.ntab) {
		return;                                            // table id not in use: the default codes stay
	}
	const uint32_t *h = w.hist + ((size_t)f * 4 + t) * 257;
	for (int i = lane; i < 320; i += 64) {
		freq[i] = i < 256 ? h[i] : (i == 256 ? 1u : 0u);   // [256]: the reserved code point (common.c:219)
		head[i] = (uint16_t)i;
		csize[i] = 0;
	}
	if (lane < 34) {
		bits[lane] = 0;
	}
	__syncthreads();
	// ---- Figure K.1
	for (;;) {
		int v[2];
#pragma unroll
		for (int pick = 0; pick < 2; ++pick) {
			unsigned long long best = ~0ull;               // (frequency << 16) | (0xffff - index): least frequent, then largest index
#pragma unroll
			for (int j = 0; j < 5; ++j) {
				const int i = lane + 64 * j;
				const uint32_t fr = freq[i];
				if (fr != 0 && !(pick == 1 && i == v[0])) {
					const unsigned long long key = ((unsigned long long)fr << 16) | (unsigned)(0xffff - i);
					best = key < best ? key : best;
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const unsigned long long other = __shfl_xor(best, o, 64);
				best = other < best ? other : best;
			}
			v[pick] = best == ~0ull ? -1 : 0xffff - (int)(best & 0xffffu);
		}
		if (v[1] < 0) {
			break;
		}
		const int v1 = v[0], v2 = v[1];
#pragma unroll
		for (int j = 0; j < 5; ++j) {
			const int i = lane + 64 * j;
			const int hd = head[i];
			if (i <= 256 && (hd == v1 || hd == v2)) {
				csize[i] = (uint8_t)min((int)csize[i] + 1, 255);
				head[i] = (uint16_t)v1;
			}
		}
		if (lane == 0) {
			freq[v1] += freq[v2];
			freq[v2] = 0;
		}
		__syncthreads();
	}
	// ---- Figure K.2: how many codes of every size
	for (int i = lane; i <= 256; i += 64) {
		const int cs = csize[i];
		if (cs > 0 && cs <= 32) {
			atomicAdd(&bits[cs], 1);
		}
	}
	__syncthreads();
	// ---- Figure K.3: no code longer than 16 bits; then the reserved code point goes
	if (lane == 0) {
		for (int i = 32; i > 16;) {
			if (bits[i] > 0) {
				int j = i - 1;
				do {
					j--;
				} while (j > 0 && bits[j] <= 0);
				bits[i] -= 2;
				bits[i - 1] += 1;
				bits[j + 1] += 2;
				bits[j] -= 1;
			} else {
				i--;
			}
		}
		int top = 16;
		while (top > 0 && bits[top] == 0) {
			top--;
		}
		if (top > 0) {
			bits[top]--;
		}
		// Annex C: first index and first code of every length (Figures C.1, C.2)
		int k = 0;
		uint32_t code = 0;
		for (int len = 1; len <= 16; ++len) {
			first_idx[len] = k;
			first_code[len] = code;
			k += bits[len];
			code = (code + (uint32_t)bits[len]) << 1;
		}
		first_idx[17] = k;
	}
	// ---- Figure K.4: symbols by (size, value); sizes before the limiting count, as in the reference
	int k = 0;
	for (int sz = 1; sz <= 32; ++sz) {
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int sym = lane + 64 * j;
			const bool hit = csize[sym] == sz;
			const unsigned long long m = __ballot(hit);
			if (hit) {
				const int pos = k + __popcll(m & ((1ull << lane) - 1ull));
				if (pos < 256) {
					outv[pos] = (uint8_t)sym;
				}
			}
			k += __popcll(m);
		}
	}
	__syncthreads();
	const int mt = min(first_idx[17], 256);
	JhDevSpec *spec = w.specs + (size_t)f * 4 + t;
	if (lane < 16) {
		spec->L[lane] = (uint8_t)bits[lane + 1];
	}
	for (int i = lane; i < 256; i += 64) {
		spec->V[i] = i < mt ? outv[i] : 0;
	}
	if (lane == 0) {
		spec->mt = (uint32_t)mt;
	}
	// ---- codes for the packer: (code << 8) | size, 0 = symbol absent
	JhHuffEnc *enc = w.codes + (size_t)f * 4 + t;
	for (int i = lane; i < 256; i += 64) {
		enc->code[i] = 0;
	}
	__syncthreads();
	for (int p = lane; p < mt; p += 64) {
		int len = 1;
		while (len < 16 && p >= first_idx[len + 1]) {
			len++;
		}
		enc->code[outv[p]] = ((first_code[len] + (uint32_t)(p - first_idx[len])) << 8) | (uint32_t)len;
	}
}

// produce_codestream's header part (encoder.c:589-645) for the adapted tables: the bytes in front of the
// DHT segments and behind them are the same for every frame of the batch.
__global__ __launch_bounds__(256) void k_enc_headers(const EncArgs a)
{
	__shared__ uint32_t seg_off[5];
	const JhEncodeWork &w = a.w;
	const int f = blockIdx.x, tid = threadIdx.x;
	const JhDevSpec *specs = w.specs + (size_t)f * 4;
	uint8_t *out = w.out + (size_t)f * w.out_stride;
	// DHT order: DC then AC for table id 0, then for id 1 (encoder.c:620-630); tables are stored [Tc*2+Th]
	const int order[4] = {0, 2, 1, 3};
	const int nseg = 2 * w.ntab;
	if (tid == 0) {
		uint32_t o = (uint32_t)w.hdr_pre_len;
		for (int i = 0; i < nseg; ++i) {
			seg_off[i] = o;
			o += 2u + 2u + 17u + specs[order[i]].mt;
		}
		seg_off[4] = o;
		w.hdr_len[f] = o + (uint32_t)w.hdr_suf_len;
	}
	__syncthreads();
	const uint32_t dht_end = seg_off[4];
	for (int i = tid; i < w.hdr_pre_len; i += 256) {
		out[i] = w.hdr_pre[i];
	}
	for (int i = tid; i < w.hdr_suf_len; i += 256) {
		out[dht_end + i] = w.hdr_suf[i];
	}
	for (int sg = 0; sg < nseg; ++sg) {
		const int t = order[sg];
		const JhDevSpec &sp = specs[t];
		uint8_t *o = out + seg_off[sg];
		const uint32_t len = 2u + 17u + sp.mt;
		for (uint32_t i = tid; i < 4u + 17u + sp.mt; i += 256) {
			uint8_t b;
			if (i == 0) {
				b = 0xff;
			} else if (i == 1) {
				b = 0xc4;
			} else if (i == 2) {
				b = (uint8_t)(len >> 8);
			} else if (i == 3) {
				b = (uint8_t)len;
			} else if (i == 4) {
				b = (uint8_t)(((t >> 1) << 4) | (t & 1));      // Tc << 4 | Th
			} else if (i < 21) {
				b = sp.L[i - 5];
			} else {
				b = sp.V[i - 21];
			}
			o[i] = b;
		}
	}
}

// the headers alone, from the tables w.specs holds (callers that fill the specs themselves)
int jh_launch_enc_headers(const JhEncodeWork &w, const JhGeom &g, hipStream_t s)
{
	EncArgs a;
	a.w = w;
	a.g = g;
	hipLaunchKernelGGL(k_enc_headers, dim3(w.nframes), dim3(256), 0, s, a);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

int jh_launch_enc_adapt(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th)
{
	EncArgs a;
	a.w = w;
	a.g = g;
	JH_T_BEGIN(th, "k_enc_adapt", s);
	hipLaunchKernelGGL(k_enc_adapt, dim3(4, w.nframes), dim3(64), 0, s, a);
	hipLaunchKernelGGL(k_enc_headers, dim3(w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------- bit lengths

__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned *scratch)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		v += __shfl_down(v, o, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		scratch[threadIdx.x >> 6] = v;
	}
	__syncthreads();
	const unsigned r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
	__syncthreads();
	return r;
}

__device__ __forceinline__ unsigned block_excl(unsigned v, unsigned *scratch, unsigned *total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned t = __shfl_up(inc, o, 64);
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

__global__ __launch_bounds__(256) void k_enc_bitlen(const EncArgs a)
{
	__shared__ uint32_t codes[4][256];
	__shared__ unsigned scratch[4];
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.y;
	for (int i = threadIdx.x; i < 4 * 256; i += 256) {
		(&codes[0][0])[i] = w.codes[(size_t)f * 4 + (i >> 8)].code[i & 255];
	}
	__syncthreads();
	const int blk = blockIdx.x * 256 + threadIdx.x;
	const int16_t *fc = w.coefs + (size_t)f * g.nblk * 64;
	unsigned bits = 0, err = 0;
	if (blk < g.nblk) {
		BlockRegs b;
		load_block(fc + (size_t)blk * 64, b);
		const int comp = g.blk_comp[blk % g.B];
		const int diff = coef_at(b, 0) - dc_predictor(fc, blk, g, w.restart_interval);
		if (diff < -2047 || diff > 2047) {
			err |= JH_EST_DC_RANGE;                        // assert (encoder.c:447)
		}
		const uint32_t *cd = codes[0 * 2 + (g.Td[comp] & 1)];
		const uint32_t *ca = codes[1 * 2 + (g.Ta[comp] & 1)];
		const int dcat = category(diff) & 15;
		const unsigned ds = cd[dcat] & 0xffu;
		if (ds == 0) {
			err |= JH_EST_NO_CODE;                         // value_to_vlc() == -1 (huffman.c:252)
		}
		bits = ds + (unsigned)dcat;
		walk_ac(b, [&](int rs, int cat, unsigned) {
			const unsigned s = ca[rs] & 0xffu;
			if (s == 0) {
				err |= JH_EST_NO_CODE;
			}
			bits += s + (unsigned)cat;
		});
		w.bitlen[(size_t)f * g.nblk + blk] = (uint16_t)bits;
		w.nzmap[(size_t)f * g.nblk + blk] = block_nzmap(b);
		w.acdc[(size_t)f * g.nblk + blk] = (uint32_t)coef_at(b, 0) << 16;
	}
	if (err) {
		atomicOr(&w.state[f].status, err);
		// inside a block the range test comes first (encoder.c:447 before write_block)
		atomicMax(&w.state[f].first_err, ~((uint32_t)blk * 2u + ((err & JH_EST_DC_RANGE) ? 0u : 1u)));
	}
	const unsigned total = block_sum(bits, scratch);
	if (threadIdx.x == 0) {
		w.wg_bits[(size_t)f * (w.nwg + 1) + blockIdx.x] = total;
	}
}

// per block k of an MCU: component | tables | where the DC predictor of the block is (encoder.c:443-458)
#define KT_COMP(t) ((t) & 3u)
#define KT_TD(t) (((t) >> 2) & 1u)
#define KT_TA(t) (((t) >> 3) & 1u)
#define KT_INNER(t) (((t) >> 4) & 1u)     // the predictor is the previous block of the same MCU
#define KT_LASTK(t) (((t) >> 5) & 31u)    // k == B - 1
#define KT_BACK(t) ((t) >> 16)            // blocks back to the predictor

__device__ __forceinline__ uint32_t make_ktab(const JhGeom &g, int k)
{
	const int comp = g.blk_comp[k];
	const int firstk = g.comp_first[comp];
	const int lastk = firstk + g.H[comp] * g.V[comp] - 1;
	const bool inner = k > firstk;
	const int back = inner ? 1 : k + g.B - lastk;
	return (uint32_t)comp | ((uint32_t)(g.Td[comp] & 1) << 2) | ((uint32_t)(g.Ta[comp] & 1) << 3) |
	       ((uint32_t)inner << 4) | ((uint32_t)(k == g.B - 1) << 5) | ((uint32_t)back << 16);
}

// Default tables: the AC part of every block's length is in JhBlkInfo already (pixel kernel).  The DC symbol
// (write_dc, coeffs.c:110-126) needs the previous block of the component (encoder.c:443-458): added here,
// with the sums per group of 256 blocks and the error conditions, from 4 bytes per block instead of 128.
// A WAVE per group, four blocks per lane: no barrier behind the first, a quarter of the workgroups (the kernel
// waits -- for its loads and for its own launch -- far more than it computes: 0.16 -> 0.08 ms per 128 metric frames).
#define DCB_GROUPS 4                                  // groups of 256 blocks per workgroup = waves
__global__ __launch_bounds__(64 * DCB_GROUPS) void k_enc_dcbits(const EncArgs a)
{
	__shared__ uint8_t dcsz[2][16];
	__shared__ uint32_t ktab[JH_MAX_MCU_BLOCKS];
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.y;
	if (threadIdx.x < 32) {
		dcsz[threadIdx.x >> 4][threadIdx.x & 15] = (uint8_t)(w.codes[(size_t)f * 4 + (threadIdx.x >> 4)].code[threadIdx.x & 15] & 0xffu);
	}
	if (threadIdx.x >= 64 && threadIdx.x < 64 + g.B) {
		ktab[threadIdx.x - 64] = make_ktab(g, threadIdx.x - 64);
	}
	__syncthreads();
	const int grp = blockIdx.x * DCB_GROUPS + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (grp >= w.nwg) {
		return;
	}
	const uint32_t *acdc = w.acdc + (size_t)f * g.nblk;
	const int ri = w.restart_interval;
	const unsigned B = (unsigned)g.B;
	const unsigned magic = 0xffffffffu / B + 1u;          // blk / B = umulhi(blk, magic) for blk < 2^28 (1 < B <= 10)
	unsigned bits = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int blk = grp * 256 + j * 64 + lane;
		if (blk < g.nblk) {
			const uint32_t r = acdc[blk];
			const unsigned acbits = r & 0xffffu;
			const int mcu = B == 1 ? blk : (int)__umulhi((unsigned)blk, magic), k = blk - mcu * (int)B;
			const uint32_t kt = ktab[k];
			int pred = 0;
			if (KT_INNER(kt) || (mcu != 0 && !(ri > 0 && mcu % ri == 0))) {
				pred = (int)acdc[blk - (int)KT_BACK(kt)] >> 16;
			}
			const int diff = ((int)r >> 16) - pred;
			unsigned e = 0;
			if (diff < -2047 || diff > 2047) {
				e |= JH_EST_DC_RANGE;                      // assert (encoder.c:447)
			}
			const int dcat = category(diff) & 15;
			const unsigned ds = dcsz[KT_TD(kt)][dcat];
			if (ds == 0 || acbits == JH_ACBITS_NOCODE) {
				e |= JH_EST_NO_CODE;                       // value_to_vlc() == -1 (huffman.c:252)
			}
			const unsigned mine = ds + (unsigned)dcat + (acbits == JH_ACBITS_NOCODE ? 0u : acbits);
			w.bitlen[(size_t)f * g.nblk + blk] = (uint16_t)mine;
			bits += mine;
			if (e) {
				atomicOr(&w.state[f].status, e);
				// inside a block the range test comes first (encoder.c:447 before write_block)
				atomicMax(&w.state[f].first_err, ~((uint32_t)blk * 2u + ((e & JH_EST_DC_RANGE) ? 0u : 1u)));
			}
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		bits += __shfl_down(bits, o, 64);
	}
	if (lane == 0) {
		w.wg_bits[(size_t)f * (w.nwg + 1) + grp] = bits;
	}
}

// one workgroup per frame: exclusive scan of the workgroup sums, total, padding, boundary words
__global__ __launch_bounds__(256) void k_enc_wg_scan(const EncArgs a)
{
	__shared__ unsigned scratch[4];
	const JhEncodeWork &w = a.w;
	const int f = blockIdx.x;
	uint32_t *wg = w.wg_bits + (size_t)f * (w.nwg + 1);
	uint32_t *us = w.us + (size_t)f * (w.us_stride / 4);
	const uint32_t cap_bits = (uint32_t)((w.us_stride - 8) * 8);
	unsigned carry = 0;
	bool over = false;
	for (int c0 = 0; c0 < w.nwg; c0 += 256) {
		const int c = c0 + threadIdx.x;
		const unsigned v = c < w.nwg ? wg[c] : 0u;
		unsigned total;
		const unsigned e = carry + block_excl(v, scratch, &total);
		if (c < w.nwg) {
			wg[c] = e;
			if ((unsigned long long)e + v > cap_bits) {
				over = true;
			} else {
				us[e >> 5] = 0;                            // word shared with the previous workgroup
			}
		}
		if ((unsigned long long)carry + total > cap_bits) {
			over = true;
		}
		carry += total;
	}
	if (over) {
		atomicOr(&w.state[f].status, JH_EST_OVERFLOW);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		JhEncState &st = w.state[f];
		const unsigned T = carry;
		wg[w.nwg] = T;
		st.total_bits = T;
		st.us_bytes = (T + 7) >> 3;
		if (!(st.status & JH_EST_OVERFLOW) && w.restart_interval == 0) {
			// flush_bits (io.c:65-87): pad the last byte with 1s
			const unsigned pad = (8 - (T & 7)) & 7;
			const unsigned mask = pad ? (((1u << pad) - 1u) << (32 - (T & 31) - pad)) : 0u;
			us[T >> 5] = __builtin_bswap32(mask);
		}
	}
}

// ---------------------------------------------------------------------------------------- restart intervals

__device__ __forceinline__ int interval_of_block(int blk, int B, int ri)
{
	return (blk / B) / ri;
}

// one thread per interval: entropy-coded bits before its end (no padding yet)
__global__ __launch_bounds__(256) void k_enc_ivl_ends(const EncArgs a)
{
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.y;
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= w.nivl) {
		return;
	}
	const int end_mcu = min((k + 1) * w.restart_interval, g.nmcu);
	const int e = end_mcu * g.B;                                   // first block AFTER the interval
	const uint32_t *wg = w.wg_bits + (size_t)f * (w.nwg + 1);
	const uint16_t *bl = w.bitlen + (size_t)f * g.nblk;
	uint32_t bits = wg[e >> 8];
	for (int i = e & ~255; i < e; ++i) {
		bits += bl[i];
	}
	w.ivl_raw[(size_t)f * w.nivl + k] = bits;
}

// one workgroup per frame: paddings, their exclusive prefix, interval ends in bytes, totals
__global__ __launch_bounds__(256) void k_enc_ivl_scan(const EncArgs a)
{
	__shared__ unsigned scratch[4];
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.x;
	JhEncState &st = w.state[f];
	if (st.status & JH_EST_OVERFLOW) {
		return;
	}
	const uint32_t *raw = w.ivl_raw + (size_t)f * w.nivl;
	uint32_t *padsum = w.ivl_padsum + (size_t)f * (w.nivl + 1);
	uint32_t *eb = w.ivl_eb + (size_t)f * w.nivl;
	const uint32_t *wg = w.wg_bits + (size_t)f * (w.nwg + 1);
	uint32_t *us = w.us + (size_t)f * (w.us_stride / 4);
	unsigned carry = 0;
	for (int k0 = 0; k0 < w.nivl; k0 += 256) {
		const int k = k0 + threadIdx.x;
		unsigned pad = 0, end = 0;
		if (k < w.nivl) {
			end = raw[k];
			const unsigned begin = k ? raw[k - 1] : 0u;
			pad = (8u - ((end - begin) & 7u)) & 7u;               // flush_bits (io.c:65-87) per interval
		}
		unsigned total;
		const unsigned before = carry + block_excl(pad, scratch, &total);
		if (k < w.nivl) {
			padsum[k] = before;
			eb[k] = (end + before + pad) >> 3;
		}
		carry += total;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		padsum[w.nivl] = carry;
		const unsigned T = wg[w.nwg] + carry;
		st.total_bits = T;
		st.us_bytes = T >> 3;
		if ((unsigned long long)T + 64 > (unsigned long long)(w.us_stride - 8) * 8) {
			st.status |= JH_EST_OVERFLOW;
		} else {
			us[T >> 5] = 0;                                        // the last, partial word is merged by OR
		}
	}
	__syncthreads();
	if (st.status & JH_EST_OVERFLOW) {
		return;
	}
	// words shared by two workgroups are merged by OR: clear them first
	for (int x = threadIdx.x; x < w.nwg; x += 256) {
		const int blk = x * 256;
		const unsigned r0 = wg[x] + padsum[interval_of_block(min(blk, g.nblk - 1), g.B, w.restart_interval)];
		us[r0 >> 5] = 0;
	}
}

// ---------------------------------------------------------------------------------------- bit packing

// what a thread reads from global memory for one group of 256 blocks
struct PackIn {
	int4 chunk[8];                    // 16-byte pieces j * 256 + tid of the group's coefficients
	uint32_t r0, r1;                  // the group's bit range
	unsigned long long nz;            // nonzero map of this thread's block
	unsigned mylen, mypad;            // bits of this thread's block (padding of its restart interval included)
	int pred;                         // DC predictor of this thread's block
	uint32_t kt;                      // its ktab entry
	bool have;
};

// ktab lives in LDS: reading it does not wait for the global loads in flight
__device__ __forceinline__ void pack_fetch(const EncArgs &a, const uint32_t *ktab, int f, int grp, int tid, PackIn &in)
{
	const JhEncodeWork &w = a.w;
	const int B = a.g.B, nblk = a.g.nblk, nmcu = a.g.nmcu;
	const uint32_t *wg = w.wg_bits + (size_t)f * (w.nwg + 1);
	const int16_t *fc = w.coefs + (size_t)f * nblk * 64;
	const int first = grp * 256, blk = first + tid;
	const int ri = w.restart_interval;
	in.have = blk < nblk;
	// per-lane loads of the (uniform) range: a scalarised load would be waited for right here
	int gi = grp;
	asm volatile("" : "+v"(gi));
	in.r0 = wg[gi];
	in.r1 = wg[gi + 1];
	const uint32_t *acdc = w.acdc + (size_t)f * nblk;
	in.mylen = in.have ? w.bitlen[(size_t)f * nblk + blk] : 0u;
	in.nz = in.have ? w.nzmap[(size_t)f * nblk + blk] : 0ull;
	const int mcu = blk / B, k = blk - mcu * B;
	const uint32_t kt = ktab[in.have ? k : 0];
	in.kt = kt;
	in.pred = 0;
	if (in.have && (KT_INNER(kt) || (mcu != 0 && !(ri > 0 && mcu % ri == 0)))) {
		in.pred = (int)acdc[blk - (int)KT_BACK(kt)] >> 16;
	}
	const int nchunk = min(256, nblk - first) * 8;
	const int4 *src = reinterpret_cast<const int4 *>(fc + (size_t)first * 64);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int ch = j * 256 + tid;
		in.chunk[j] = ch < nchunk ? src[ch] : make_int4(0, 0, 0, 0);
	}
	in.mypad = 0;
	if (ri > 0) {
		// restart intervals (extension): everything moves by the padding before its interval, and the last
		// block of an interval carries the interval's padding
		const uint32_t *padsum = w.ivl_padsum + (size_t)f * (w.nivl + 1);
		const int next = first + 256;
		in.r0 += padsum[interval_of_block(first, B, ri)];
		in.r1 += next < nblk ? padsum[interval_of_block(next, B, ri)] : padsum[w.nivl];
		if (in.have && KT_LASTK(kt) && ((mcu + 1) % ri == 0 || mcu + 1 == nmcu)) {
			const int iv = mcu / ri;
			in.mypad = padsum[iv + 1] - padsum[iv];
			in.mylen += in.mypad;
		}
	}
}

// A workgroup packs the groups blockIdx.x, blockIdx.x + gridDim.x, ... of 256 blocks.  A group is staged in
// LDS (coalesced 16-byte loads, issued while the previous group is being packed) and a 64-bit map of the
// nonzero coefficients of every block is built on the way; a thread then visits only the nonzero
// coefficients of its block (Figure F.2 without the 63-step scan: the zero run is the distance to the
// previous nonzero) and ORs its code words into an LDS image of the group's bit range.
__global__ __launch_bounds__(256) void k_enc_pack(const EncArgs a)
{
	// per table id and symbol: the code word moved up by the symbol's category -- room for the extra bits -- and the
	// bits of the two together (a look-up gives both at once; behind each table a few entries of room: a category
	// of 16, which only a frame that has failed already can have, reads there)
	__shared__ uint2 actab[2][256 + 16];
	__shared__ uint2 dctab[2][16 + 16];
	__shared__ unsigned scratch[4];
	__shared__ __align__(16) uint32_t win[WIN_WORDS + 4];      // + sink
	__shared__ uint32_t stage[256 * ST_STRIDE];
	__shared__ unsigned long long nzmap[256];
	__shared__ uint32_t bstart[256], binfo[256], hist[64];
	__shared__ int bpred[256];
	__shared__ uint8_t owner[256];
	const JhEncodeWork &w = a.w;
	const JhGeom &g = a.g;
	const int f = blockIdx.y;
	if (w.state[f].status & JH_EST_OVERFLOW) {
		return;
	}
	const int tid = threadIdx.x;
	__shared__ uint32_t ktab[JH_MAX_MCU_BLOCKS];
	if (tid < g.B) {
		ktab[tid] = make_ktab(g, tid);
	}
	__syncthreads();
	PackIn in;
	pack_fetch(a, ktab, f, blockIdx.x, tid, in);
#pragma unroll
	for (int th = 0; th < 2; ++th) {
		const unsigned cat = (unsigned)tid & 15u;
		const uint32_t e = w.codes[(size_t)f * 4 + 2 + th].code[tid];
		actab[th][tid] = make_uint2((e >> 8) << cat, (e & 0xffu) + cat);
		if (tid < 16) {
			const uint32_t d = w.codes[(size_t)f * 4 + th].code[tid];
			dctab[th][tid] = make_uint2((d >> 8) << cat, (d & 0xffu) + cat);
		}
	}
	for (int i = tid; i < WIN_WORDS / 4; i += 256) {
		reinterpret_cast<int4 *>(win)[i] = make_int4(0, 0, 0, 0);
	}
	uint32_t *us = w.us + (size_t)f * (w.us_stride / 4);

	for (int grp = blockIdx.x; grp < w.nwg; grp += gridDim.x) {
		// ---- stage the blocks, build the nonzero maps
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int ch = j * 256 + tid;
			const int4 v = in.chunk[j];
			uint32_t *row = stage + (ch >> 3) * ST_STRIDE + (ch & 7) * 4;
			row[0] = (uint32_t)v.x;
			row[1] = (uint32_t)v.y;
			row[2] = (uint32_t)v.z;
			row[3] = (uint32_t)v.w;
		}
		nzmap[tid] = in.nz;
		const uint32_t r0 = __builtin_amdgcn_readfirstlane(in.r0), r1 = __builtin_amdgcn_readfirstlane(in.r1);
		const unsigned mylen = in.mylen, mypad = in.mypad;
		const int pred = in.pred;
		const uint32_t kt = in.kt;
		const bool have = in.have;
		// the next group's reads are in flight while this one is packed
		if (grp + (int)gridDim.x < w.nwg) {
			pack_fetch(a, ktab, f, grp + (int)gridDim.x, tid, in);
		}
		binfo[tid] = mylen | (mypad << 16) | (KT_TD(kt) << 20) | (KT_TA(kt) << 21) | ((uint32_t)have << 22);
		bpred[tid] = pred;
		if (tid < 64) {
			hist[tid] = 0;
		}
		unsigned total;
		bstart[tid] = r0 + block_excl(mylen, scratch, &total);            // barriers inside: LDS contents visible after
		// ---- blocks change hands so that the blocks of a wave have similar numbers of nonzero coefficients
		// (the wave walks as long as its busiest lane): counting sort by that number, busiest first
		const int bin = 63 - __popcll(nzmap[tid] & ~1ull);
		const unsigned rank = atomicAdd(&hist[bin], 1u);
		__syncthreads();
		{
			const int lane = tid & 63;
			const unsigned h = hist[lane];
			unsigned inc = h;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned t = __shfl_up(inc, o, 64);
				inc += lane >= o ? t : 0u;
			}
			owner[__shfl(inc - h, bin, 64) + rank] = (uint8_t)tid;
		}
		__syncthreads();
		const int lb = owner[tid];                                         // the block this thread packs
		const uint32_t start = bstart[lb], inf = binfo[lb];
		const unsigned plen = inf & 0xffffu, ppad = (inf >> 16) & 15u;
		const bool phave = (inf >> 22) & 1u;
		const int16_t *mine = reinterpret_cast<const int16_t *>(stage + lb * ST_STRIDE);
		const unsigned long long nzac = nzmap[lb] & ~1ull;
		const int diff = phave ? (int)mine[0] - bpred[lb] : 0;
		const uint2 *cd = dctab[(inf >> 20) & 1u];
		const uint2 *ca = actab[(inf >> 21) & 1u];

		const uint32_t wfirst = r0 >> 5, wlast = r1 > r0 ? (r1 - 1) >> 5 : wfirst;
		for (uint32_t wbase = wfirst; wbase <= wlast && r1 > r0; wbase += WIN_WORDS) {
			// does my bit range [start, start + mylen) touch this window?
			const uint32_t lo_bit = wbase << 5;
			const unsigned long long hi_bit = ((unsigned long long)wbase + WIN_WORDS) << 5;
			if (phave && plen && (unsigned long long)start + plen > lo_bit && start < hi_bit) {
				// a symbol (at most 31 bits) is ORed into the two words it can touch; words outside this
				// pass's window go to the sink behind it.  No accumulator, no branch.
				// (positions count from the window's first bit -- a multiple of 32, and a block is never more than the
				// group's bits away from it: a signed number of bits.  The word's byte offset is taken with an arithmetic
				// shift, so that a word in front of the window is a huge unsigned number and goes to the sink, while the
				// word behind the last one in front is 0: a symbol across the window's start leaves its tail there.)
				int pos = (int)(start - lo_bit);
				auto put = [&](unsigned code, unsigned len) {
					const unsigned long long v = (unsigned long long)code << ((64u - ((unsigned)pos & 31u) - len) & 63u);
					const uint32_t at = (uint32_t)(pos >> 3) & ~3u;
					unsigned char *w8 = reinterpret_cast<unsigned char *>(win);
					atomicOr(reinterpret_cast<uint32_t *>(w8 + min(at, (uint32_t)WIN_WORDS * 4u)), (uint32_t)(v >> 32));
					atomicOr(reinterpret_cast<uint32_t *>(w8 + min(at + 4u, (uint32_t)WIN_WORDS * 4u)), (uint32_t)v);
					pos += (int)len;
				};
				// encode_cat / encode_extra (coeffs.c:56-83) in one go: a negative value's extra bits are those of
				// value - 1, and the category is what is left of 32 bits behind the leading sign bits of THAT
				const int ds = diff + (diff >> 31);
				const int dcat = diff ? 32 - sign_bits(ds) : 0;
				const uint2 dcode = cd[dcat];                  // write_dc (coeffs.c:110-126)
				put(dcode.x | __builtin_amdgcn_ubfe((unsigned)ds, 0u, (unsigned)dcat), dcode.y);
				// write_ac (coeffs.c:164-194); the next nonzero coefficient is read while this one is emitted
				// (behind the last nonzero coefficient, coefficient 63 is read for nothing.  ORing bit 63 into the map
				// instead of the select costs 3 %, and so does walking the map as two 32-bit halves: measured)
				unsigned long long m = nzac;
				int prev = 0;
				int k = m ? __builtin_ctzll(m) : 63;
				int c = mine[k];
				while (m) {
					m &= m - 1;
					const int kn = m ? __builtin_ctzll(m) : 63;
					const int cn = mine[kn];
					const int r = k - prev - 1;
					prev = k;
					if (r > 15) {
						const uint2 z = ca[0xf0];              // ZRL
						for (int q = r >> 4; q > 0; --q) {
							put(z.x, z.y);
						}
					}
					const int cs = c + (c >> 31);              // c is not zero here
					const int cat = 32 - sign_bits(cs);
					const uint2 e = ca[((r & 15) << 4) + cat];
					put(e.x | __builtin_amdgcn_ubfe((unsigned)cs, 0u, (unsigned)cat), e.y);
					k = kn;
					c = cn;
				}
				if (prev != 63) {
					const uint2 e = ca[0];                     // EOB
					put(e.x, e.y);
				}
				if (ppad) {
					put((1u << ppad) - 1u, ppad);              // flush_bits at the end of a restart interval
				}
			}
			__syncthreads();
			// store the image: whole words plainly, the (at most two) words shared with neighbours by OR; the
			// image is left zero for the next pass
			const uint32_t nw = min((uint32_t)WIN_WORDS, wlast - wbase + 1);
			for (uint32_t i = tid; i < nw; i += 256) {
				const uint32_t wi = wbase + i;
				const uint32_t v = __builtin_bswap32(win[i]);  // MSB-first bit stream -> byte order
				win[i] = 0;
				const bool whole = (wi << 5) >= r0 && ((unsigned long long)(wi + 1) << 5) <= r1;
				if (whole) {
					us[wi] = v;
				} else if (v) {
					atomicOr(&us[wi], v);
				}
			}
			__syncthreads();
		}
		__syncthreads();                                   // stage[] and nzmap[] are rewritten next
	}
}

// ---------------------------------------------------------------------------------------- byte stuffing

// RSTn markers go in FRONT of the bytes at eb[0 .. n): how many of them fall into [lo, hi); *first = index
// of the first one at or after lo
__device__ __forceinline__ unsigned markers_in(const uint32_t *eb, int n, uint32_t lo, uint32_t hi, int *first)
{
	int a = 0, b = n;                                             // lower bound of lo
	while (a < b) {
		const int m = (a + b) >> 1;
		if (eb[m] < lo) {
			a = m + 1;
		} else {
			b = m;
		}
	}
	if (first) {
		*first = a;
	}
	unsigned c = 0;
	while (a < n && eb[a] < hi) {
		++a;
		++c;
	}
	return c;
}

// The stream of a frame is sized for the worst case but mostly a tenth of that long, and how long is known on
// the device only: a fixed number of workgroups per frame take the chunks in turn.
__global__ __launch_bounds__(256) void k_enc_ffcount(const EncArgs a)
{
	__shared__ unsigned scratch[4];
	const JhEncodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhEncState st = w.state[f];
	const uint32_t nbytes = (st.status & JH_EST_OVERFLOW) ? 0u : st.us_bytes;
	for (uint32_t chunk = blockIdx.x; chunk * (uint32_t)FF_CHUNK < nbytes; chunk += gridDim.x) {
		const uint32_t base = chunk * FF_CHUNK + threadIdx.x * 16;
		unsigned cnt = 0;
		if (base < nbytes) {
			const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(w.us) + (size_t)f * w.us_stride + base);
			const unsigned ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const unsigned byte = (ws[i >> 2] >> (8 * (i & 3))) & 0xffu;
				cnt += (base + i < nbytes && byte == 0xffu) ? 1u : 0u;
			}
			if (w.restart_interval > 0) {
				cnt += 2u * markers_in(w.ivl_eb + (size_t)f * w.nivl, w.nivl - 1, base, base + 16, nullptr);
			}
		}
		const unsigned total = block_sum(cnt, scratch);
		if (threadIdx.x == 0) {
			w.chunk_ff[(size_t)f * w.nchunk + chunk] = total;
		}
	}
}

__global__ __launch_bounds__(256) void k_enc_ff_scan(const EncArgs a)
{
	__shared__ unsigned scratch[4];
	const JhEncodeWork &w = a.w;
	const int f = blockIdx.x;
	JhEncState &st = w.state[f];
	const uint32_t nbytes = (st.status & JH_EST_OVERFLOW) ? 0u : st.us_bytes;
	const int nch = (int)((nbytes + FF_CHUNK - 1) / FF_CHUNK);
	uint32_t *ff = w.chunk_ff + (size_t)f * w.nchunk;
	unsigned carry = 0;
	for (int c0 = 0; c0 < nch; c0 += 256) {
		const int c = c0 + threadIdx.x;
		const unsigned v = c < nch ? ff[c] : 0u;
		unsigned total;
		const unsigned e = carry + block_excl(v, scratch, &total);
		if (c < nch) {
			ff[c] = e;
		}
		carry += total;
	}
	if (threadIdx.x == 0) {
		const uint32_t ecs = nbytes + carry;
		st.ecs_bytes = ecs;
		uint8_t *out = w.out + (size_t)f * w.out_stride + w.hdr_len[f];
		if ((size_t)w.hdr_len[f] + ecs + 2 <= w.out_stride) {
			out[ecs] = 0xff;                               // EOI (encoder.c:402-410)
			out[ecs + 1] = 0xd9;
		} else {
			st.status |= JH_EST_OVERFLOW;
		}
	}
}

// A chunk's output is put together in LDS (at the byte phase its place in the output has) and stored as whole
// dwords, the ragged ends byte by byte.  Worst case per chunk: every byte 0xFF and a restart marker in front of
// every byte.
#define STUFF_LDS (FF_CHUNK * 4 + 16)
__global__ __launch_bounds__(256) void k_enc_stuff(const EncArgs a)
{
	__shared__ unsigned scratch[4];
	__shared__ __align__(16) uint8_t obuf[STUFF_LDS];
	const JhEncodeWork &w = a.w;
	const int f = blockIdx.y;
	const JhEncState st = w.state[f];
	if (st.status & JH_EST_OVERFLOW) {
		return;
	}
	const uint32_t nbytes = st.us_bytes;
	const uint32_t *eb = w.ivl_eb + (size_t)f * w.nivl;
	const int nmark = w.restart_interval > 0 ? w.nivl - 1 : 0;
	uint8_t *out = w.out + (size_t)f * w.out_stride + w.hdr_len[f];
	for (uint32_t chunk = blockIdx.x; chunk * (uint32_t)FF_CHUNK < nbytes; chunk += gridDim.x) {
		const uint32_t cbase = chunk * FF_CHUNK, base = cbase + threadIdx.x * 16;
		unsigned bytes[16];
		unsigned cnt = 0;
		int kb = 0;                                                   // next restart marker
		if (base < nbytes) {
			const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(w.us) + (size_t)f * w.us_stride + base);
			const unsigned ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				bytes[i] = (ws[i >> 2] >> (8 * (i & 3))) & 0xffu;
				cnt += (base + i < nbytes && bytes[i] == 0xffu) ? 1u : 0u;
			}
			if (nmark > 0) {
				cnt += 2u * markers_in(eb, nmark, base, base + 16, &kb);
			}
		}
		unsigned total;
		const unsigned within = block_excl(cnt, scratch, &total);       // barriers inside: obuf is free again
		const uint32_t first = cbase + w.chunk_ff[(size_t)f * w.nchunk + chunk];   // where the chunk's output starts
		const uint32_t phase = (uint32_t)(reinterpret_cast<uintptr_t>(out + first) & 3u);
		if (base < nbytes) {
			uint32_t o = phase + threadIdx.x * 16 + within;        // write_ecs_byte (io.c:277-290)
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				if (base + i < nbytes) {
					if (kb < nmark && eb[kb] == base + i) {
						obuf[o++] = 0xff;                              // RSTn in front of the interval's first byte
						obuf[o++] = (uint8_t)(0xd0 + (kb & 7));
						++kb;
					}
					obuf[o++] = (uint8_t)bytes[i];
					if (bytes[i] == 0xffu) {
						obuf[o++] = 0x00;
					}
				}
			}
		}
		__syncthreads();
		const uint32_t nout = min((uint32_t)FF_CHUNK, nbytes - cbase) + total;
		uint8_t *dst = out + first - phase;                        // dword aligned
		const uint32_t end = phase + nout;
		for (uint32_t x = threadIdx.x * 4; x < end; x += 1024) {
			if (x >= phase && x + 4 <= end) {
				*reinterpret_cast<uint32_t *>(dst + x) = *reinterpret_cast<const uint32_t *>(obuf + x);
			} else {
				for (uint32_t i = x < phase ? phase : x; i < x + 4 && i < end; ++i) {
					dst[i] = obuf[i];
				}
			}
		}
		__syncthreads();
	}
}

int jh_launch_enc_entropy(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th, bool acbits_ready)
{
	EncArgs a;
	a.w = w;
	a.g = g;
	if (acbits_ready) {
		JH_T_BEGIN(th, "k_enc_dcbits", s);
		hipLaunchKernelGGL(k_enc_dcbits, dim3((w.nwg + DCB_GROUPS - 1) / DCB_GROUPS, w.nframes), dim3(64 * DCB_GROUPS), 0, s, a);
		JH_T_END(th, s);
	} else {
		JH_T_BEGIN(th, "k_enc_bitlen", s);
		hipLaunchKernelGGL(k_enc_bitlen, dim3(w.nwg, w.nframes), dim3(256), 0, s, a);
		JH_T_END(th, s);
	}
	JH_T_BEGIN(th, "k_enc_wg_scan", s);
	hipLaunchKernelGGL(k_enc_wg_scan, dim3(w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	if (w.restart_interval > 0) {
		JH_T_BEGIN(th, "k_enc_ivl", s);
		hipLaunchKernelGGL(k_enc_ivl_ends, dim3((w.nivl + 255) / 256, w.nframes), dim3(256), 0, s, a);
		hipLaunchKernelGGL(k_enc_ivl_scan, dim3(w.nframes), dim3(256), 0, s, a);
		JH_T_END(th, s);
	}
	JH_T_BEGIN(th, "k_enc_pack", s);
	// a workgroup packs several groups of 256 blocks in turn when the frame has plenty of them
	const int pack_wgs = w.nwg >= 64 ? (w.nwg + 3) / 4 : w.nwg;
	hipLaunchKernelGGL(k_enc_pack, dim3(pack_wgs, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	// chunks per frame are a capacity (the stream's real length is on the device): some workgroups per frame
	// walk them, about 8192 in all
	const int ff_wgs = min(w.nchunk, max(32, 8192 / w.nframes));
	JH_T_BEGIN(th, "k_enc_ffcount", s);
	hipLaunchKernelGGL(k_enc_ffcount, dim3(ff_wgs, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_enc_ff_scan", s);
	hipLaunchKernelGGL(k_enc_ff_scan, dim3(w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	JH_T_BEGIN(th, "k_enc_stuff", s);
	hipLaunchKernelGGL(k_enc_stuff, dim3(ff_wgs, w.nframes), dim3(256), 0, s, a);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================== Motion JPEG stream out
//
// The reference writes its one codestream to a FILE* as it goes (produce_codestream, encoder.c:589-645, through
// write_ecs_byte, io.c:277-290).  A batch of frames leaves the device as ONE run of bytes instead: the finished
// codestreams (headers, entropy-coded data, EOI), back to back in frame order, with a table of offsets --
// a Motion JPEG stream as jpeg_hip_mjpeg_split() reads it.  The destination is device memory; api.hip sends it on
// to page-locked host memory by DMA when that is where the caller wants it.

// One workgroup: the frames' lengths (0 for a frame that failed), their exclusive prefix, and which frames still
// fit `cap`.  off[nframes] = the stream's length.  len[f] = JH_STREAM_NOFIT: codestream complete, no room.
__global__ __launch_bounds__(256) void k_enc_stream_offsets(const JhEncodeWork w, unsigned long long cap,
                                                            unsigned long long *off, uint32_t *len)
{
	__shared__ unsigned long long scratch[256];
	__shared__ unsigned long long carry, endpos;
	if (threadIdx.x == 0) {
		carry = 0;
		endpos = ~0ull;
	}
	__syncthreads();
	for (int f0 = 0; f0 < w.nframes; f0 += 256) {
		const int f = f0 + threadIdx.x;
		unsigned long long n = 0;
		if (f < w.nframes) {
			const JhEncState st = w.state[f];
			if (st.first_err == 0 && !(st.status & (JH_EST_DC_RANGE | JH_EST_NO_CODE | JH_EST_OVERFLOW))) {
				n = (unsigned long long)w.hdr_len[f] + st.ecs_bytes + 2u;
			}
		}
		// a frame that does not fit takes no room, and neither does any frame behind it (a stream with a hole
		// would misplace what follows): inclusive scan of the lengths, cut where it passes cap
		scratch[threadIdx.x] = n;
		__syncthreads();
		for (int d = 1; d < 256; d <<= 1) {
			const unsigned long long v = threadIdx.x >= (unsigned)d ? scratch[threadIdx.x - d] : 0;
			__syncthreads();
			scratch[threadIdx.x] += v;
			__syncthreads();
		}
		const unsigned long long incl = carry + scratch[threadIdx.x];
		const unsigned long long total = carry + scratch[255];
		__syncthreads();
		if (f < w.nframes) {
			const bool fits = incl <= cap;               // lengths are not negative: once false, false for every later frame
			off[f] = fits ? incl - n : (cap < incl - n ? cap : incl - n);
			len[f] = fits ? (uint32_t)n : (n ? JH_STREAM_NOFIT : 0u);
			if (!fits && n) {
				atomicMin(&endpos, incl - n);
			}
		}
		if (threadIdx.x == 0) {
			carry = total;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		off[w.nframes] = endpos < carry ? endpos : carry;    // the stream's length: up to the first frame that did not fit
	}
}

// Some workgroups (gridDim.x), each walking every frame and taking its share of the frame's 16-byte pieces: frame f's
// bytes from its slot to dst + off[f].  The destination is written in aligned 16-byte pieces (one per thread, 4 KB
// per workgroup step), the source read as aligned dwords and shifted into place; the up to 15 bytes in front of the
// first and behind the last whole piece go byte by byte (the neighbouring frames write the rest of those pieces).
__global__ __launch_bounds__(256) void k_enc_stream_copy(const JhEncodeWork w, uint8_t *dst, const unsigned long long *off,
                                                         const uint32_t *len)
{
	for (int f = 0; f < w.nframes; ++f) {
		const uint32_t n = len[f];
		if (n == 0 || n == JH_STREAM_NOFIT) {
			continue;
		}
		uint8_t *d = dst + off[f];
		const uint8_t *src = w.out + (size_t)f * w.out_stride;         // 4096-aligned
		const uint32_t head = min(n, (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u));
		const uint32_t npiece = (n - head) >> 4;
		const uint32_t tail0 = head + (npiece << 4);
		if ((unsigned)f % gridDim.x == blockIdx.x) {
			if (threadIdx.x < head) {
				d[threadIdx.x] = src[threadIdx.x];
			}
			if (threadIdx.x >= 32 && tail0 + (threadIdx.x - 32) < n) {
				d[tail0 + (threadIdx.x - 32)] = src[tail0 + (threadIdx.x - 32)];
			}
		}
		const uint32_t sh = head & 3u;                                 // the source's byte phase, the same for every piece
		const uint32_t *sw = reinterpret_cast<const uint32_t *>(src + (head & ~3u));
		uint4 *dw = reinterpret_cast<uint4 *>(d + head);
		for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npiece; i += gridDim.x * 256) {
			const uint32_t *p = sw + 4 * (size_t)i;
			uint4 v;
			if (sh == 0) {
				v = make_uint4(p[0], p[1], p[2], p[3]);
			} else {
				// bytes sh .. sh + 15 of five dwords; the fifth lies inside the allocation (16 spare bytes behind the
				// last slot, api.hip)
				const uint32_t a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4];
				v = make_uint4(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh),
				               __builtin_amdgcn_alignbyte(a3, a2, sh), __builtin_amdgcn_alignbyte(a4, a3, sh));
			}
			dw[i] = v;
		}
	}
}

int jh_launch_enc_stream(const JhEncodeWork &w, void *dst, size_t cap, unsigned long long *off, uint32_t *len,
                         hipStream_t s, const JhTimerHook *th)
{
	JH_T_BEGIN(th, "k_enc_stream", s);
	hipLaunchKernelGGL(k_enc_stream_offsets, dim3(1), dim3(256), 0, s, w, (unsigned long long)cap, off, len);
	static const int stream_wgs = [] {
		const char *e = getenv("JPEG_AMD_STREAM_WGS");                  // experiments
		const int v = e ? atoi(e) : 0;
		return v > 0 && v <= 8192 ? v : 1024;
	}();
	hipLaunchKernelGGL(k_enc_stream_copy, dim3(stream_wgs), dim3(256), 0, s, w, (uint8_t *)dst, off, len);
	JH_T_END(th, s);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
