// jpeg_amd/csrc/huffcore.hpp -- the per-thread Huffman decoding core shared by the synchronisation
// kernels (huffsync.hip) and the scan / write kernels (huffdec.hip).  Device code only.
#pragma once

#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"

struct DecTables {                    // workgroup-shared decode context (LDS)
	uint8_t blk_dc[JH_MAX_MCU_BLOCKS], blk_ac[JH_MAX_MCU_BLOCKS], blk_comp[JH_MAX_MCU_BLOCKS];
	uint8_t pad[2];
};

struct DecState {
	uint32_t pos;                     // bit position of the next symbol
	uint32_t seg;                     // restart segment that contains pos
	int c, z;                         // block inside the MCU, next zig-zag index (0 = DC expected)
	int reset;                        // a restart boundary was crossed
	int nb;                           // blocks completed (partial MCUs before a boundary are discarded)
	int dc[JH_MAX_COMP];              // DC difference sums since the last reset (WRITE: the predictors)
};

__device__ __forceinline__ unsigned long long pack_state(const DecState &s)
{
	return (unsigned long long)s.pos | ((unsigned long long)(s.c & 0xff) << 32) |
	       ((unsigned long long)(s.z & 0xff) << 40) | ((unsigned long long)(s.reset & 1) << 48);
}

__device__ __forceinline__ uint32_t load_be32(const uint8_t *cs, uint32_t word)
{
	return __builtin_bswap32(reinterpret_cast<const uint32_t *>(cs)[word]);
}

// F.12: extend the sign of a cat-bit magnitude field (coeffs.c:33-48)
__device__ __forceinline__ int extend(unsigned extra, int cat)
{
	return cat == 0 ? 0 : ((extra >> (cat - 1)) ? (int)extra : (int)extra - (1 << cat) + 1);
}

// Decode every symbol that STARTS before bit `stop`, beginning at st.pos.  WRITE selects the final
// pass (coefficients stored, st.dc = predictors, blk = index of the block in progress).
template <bool WRITE>
__device__ __forceinline__ void decode_span(DecState &st, const uint32_t stop, const uint8_t *cs,
                                            const uint32_t total_bits, const uint32_t *seg_start,
                                            const uint32_t nseg, const JhHuffLut *luts,
                                            const DecTables &tb, const int B,
                                            int16_t *coefs, int blk, const int nblk)
{
	if (st.pos >= stop || st.pos >= total_bits) {
		return;
	}
	uint32_t seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
	uint32_t wpos = st.pos >> 5;
	unsigned long long buf = ((unsigned long long)load_be32(cs, wpos) << 32) | load_be32(cs, wpos + 1);
	buf <<= (st.pos & 31u);
	int avail = 64 - (int)(st.pos & 31u);
	wpos += 2;
	uint32_t pos = st.pos;
	int c = st.c, z = st.z;

	while (pos < stop) {
		if (avail < 32) {
			buf |= (unsigned long long)load_be32(cs, wpos++) << (32 - avail);
			avail += 32;
		}
		const JhHuffLut &lut = luts[z == 0 ? tb.blk_dc[c] : tb.blk_ac[c]];
		unsigned e = lut.fast[(unsigned)(buf >> (64 - JH_LUT_BITS))];
		int len = (int)(e >> 8);
		unsigned sym = e & 0xffu;
		bool invalid = false;
		if (len == 0) {
			// code longer than the fast table: canonical search on the left-aligned 16-bit window
			const unsigned v = (unsigned)(buf >> 48);
			len = JH_LUT_BITS + 1;
#pragma unroll
			for (int l = JH_LUT_BITS + 1; l < 16; ++l) {
				len += v >= lut.limit[l] ? 1 : 0;
			}
			if (v >= lut.limit[16]) {
				invalid = true;       // no code matches: the reference reads on to the end of the segment
				sym = 0;
			} else {
				sym = lut.huffval[(lut.offs[len] + (int)(v >> (16 - len))) & 0xff];
			}
		}
		const int cat = (int)(sym & 15u);
		const int total = len + cat;
		if (invalid || pos + (uint32_t)total > seg_end) {
			// end of this restart segment (padding bits, truncated data or garbage): decoder.c:376-381.
			// The MCU in progress is dropped (context->mblocks is not incremented), the next segment
			// starts with block 0 / DC predictors 0 (decoder.c:369-373).
			st.nb -= c;
			if (WRITE) {
				blk -= c;
			}
			c = 0;
			z = 0;
			st.reset = 1;
#pragma unroll
			for (int k = 0; k < JH_MAX_COMP; ++k) {
				st.dc[k] = 0;
			}
			st.seg++;
			if (st.seg >= nseg) {
				pos = total_bits;
				break;
			}
			pos = seg_end;
			seg_end = min(seg_start[st.seg + 1] * 8u, total_bits);
			wpos = pos >> 5;
			buf = ((unsigned long long)load_be32(cs, wpos) << 32) | load_be32(cs, wpos + 1);
			buf <<= (pos & 31u);
			avail = 64 - (int)(pos & 31u);
			wpos += 2;
			continue;
		}
		const unsigned extra = cat ? (unsigned)((buf << len) >> (64 - cat)) : 0u;
		buf <<= total;
		avail -= total;
		pos += (uint32_t)total;
		if (z == 0) {
			const int diff = extend(extra, cat);
			const int comp = tb.blk_comp[c];
			// dynamic index into a tiny array: keep it in registers with a select chain
			int val = 0;
#pragma unroll
			for (int k = 0; k < JH_MAX_COMP; ++k) {
				if (k == comp) {
					st.dc[k] += diff;
					val = st.dc[k];
				}
			}
			if (WRITE && blk >= 0 && blk < nblk) {
				coefs[(size_t)blk * 64] = (int16_t)val;      // DC un-prediction (decoder.c:351-355)
			}
			z = 1;
		} else if (sym == 0) {
			z = 64;                                            // EOB (coeffs.c:241-243)
		} else {
			z += (int)(sym >> 4);                              // zero run (coeffs.c:246)
			if (WRITE && cat != 0 && z < 64 && blk >= 0 && blk < nblk) {
				coefs[(size_t)blk * 64 + z] = (int16_t)extend(extra, cat);
			}
			z++;
		}
		if (z >= 64) {
			z = 0;
			c = (c + 1 == B) ? 0 : c + 1;
			st.nb++;
			if (WRITE) {
				blk++;
			}
		}
	}
	st.pos = pos;
	st.c = c;
	st.z = z;
}

// segment that contains byte offset `byte` : largest k with seg_start[k] <= byte
__device__ __forceinline__ uint32_t find_segment(const uint32_t *seg_start, uint32_t nseg, uint32_t byte)
{
	uint32_t lo = 0, hi = nseg;   // invariant: seg_start[lo] <= byte, answer in [lo, hi)
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (seg_start[mid] <= byte) {
			lo = mid;
		} else {
			hi = mid;
		}
	}
	return lo;
}

__device__ __forceinline__ void load_tables(const JhDecodeWork &w, const JhDecFrame &fr, JhHuffLut *luts, DecTables *tb,
                                            const JhGeom &g)
{
	const uint32_t *src = reinterpret_cast<const uint32_t *>(w.luts + fr.lut_off);
	uint32_t *dst = reinterpret_cast<uint32_t *>(luts);
	const int nw = (int)(fr.n_lut * sizeof(JhHuffLut) / 4);
	for (int i = threadIdx.x; i < nw; i += blockDim.x) {
		dst[i] = src[i];
	}
	if (threadIdx.x < JH_MAX_MCU_BLOCKS) {
		// read through the global pointer: indexing the by-value copy would put it in scratch
		const JhDecFrame *gf = w.frames + blockIdx.y;
		tb->blk_dc[threadIdx.x] = gf->blk_dc[threadIdx.x];
		tb->blk_ac[threadIdx.x] = gf->blk_ac[threadIdx.x];
		tb->blk_comp[threadIdx.x] = (uint8_t)g.blk_comp[threadIdx.x];
	}
	__syncthreads();
}

__device__ __forceinline__ void store_info(JhSubInfo *dst, const DecState &s)
{
	JhSubInfo r;
	r.state = pack_state(s);
	r.seg = s.seg;
	r.nb = s.nb;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		r.dc[k] = s.dc[k];
	}
	*dst = r;
}

// chain state carried from one sub-sequence into the next: counters restart, position/phase stay
__device__ __forceinline__ void next_subsequence(DecState &s)
{
	s.reset = 0;
	s.nb = 0;
#pragma unroll
	for (int k = 0; k < JH_MAX_COMP; ++k) {
		s.dc[k] = 0;
	}
}

#define STATE_SYNC_MASK 0x0000ffffffffffffull      // pos, c, z (the reset flag is not part of the parse state)

struct HuffArgs {
	JhDecodeWork w;
	JhGeom g;
	int round;                         // synchronisation round (fix-up kernel)
};
