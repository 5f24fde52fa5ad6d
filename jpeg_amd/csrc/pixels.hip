// jpeg_amd/csrc/pixels.hip -- the dense half of the codec: fused pixel pipelines for gfx950.
//
//   k_decode_pixels  = epilogue() of the reference (decoder.c:456-470) in ONE kernel:
//       dequantize_block (imgproc.c:10-19)  ->  idct (imgproc.c:130-149)  ->  level shift
//       (imgproc.c:192-194)  ->  conv_blocks_to_frame (imgproc.c:233-262)  ->  replication
//       up-sampling (frame.c:38-82)  ->  frame_to_rgb (frame.c:188-244)  ->  roundf/clamp/
//       u8 or big-endian u16 (frame.c:369-400).
//   k_encode_pixels  = prologue() (encoder.c:176-193) in ONE kernel:
//       sample -> float + edge replication (frame.c:295-345)  ->  frame_to_ycc (frame.c:154-186,
//       only x < X, y < Y: quirk Q7)  ->  box-mean down-sampling (frame.c:84-132)  ->
//       conv_frame_to_blocks (imgproc.c:264-293)  ->  level shift + fdct (imgproc.c:202-230)
//       ->  quantize_block (imgproc.c:21-30).
//
// Bit-exactness: this translation unit is compiled with -ffp-contract=off (no FMA), the DCT
// uses the literal table T1 and the reference's summation order (index ascending), colour maths
// runs in binary64 exactly where the reference's C expressions promote to double, and fp32
// division is hipcc's correctly rounded default.  MFMA is deliberately not used: its fused
// accumulation cannot reproduce the reference's roundings.
//
// The reference starts every 8-term sum with `s = 0.f; s += p0`.  `0.f + p0` differs from `p0`
// only for p0 == -0.f, and a different SIGN OF ZERO in a partial sum can only change the sign of a
// zero in the transform's result; the next consumer is either `+ shift` (decode) or
// `(int)roundf(x / q)` (encode), both blind to it.  The sums here therefore start at p0.
//
// Work decomposition.  One workgroup = a run of G MCUs of one MCU row (about 96 blocks); the
// tile lives in LDS as one float plane per component AT COMPONENT RESOLUTION, 8*V rows of
// G*8*H (+4) floats.  All stages are "tasks" strided over the 256 threads:
//   decode: A  (block, zig-zag octet): one coalesced 16-byte load, parked as it is in LDS (the int16 lines
//              of the tile's blocks lie over the float planes, which are not in use yet)
//           B  (block, row)   : the row's 8 coefficients picked out of the block's zig-zag line, dequantised,
//              1-D inverse transform of the row in 16 VGPRs (packed f32 maths); written to the planes
//              once every thread has read its coefficients
//           C  (block, column): 1-D inverse transform of a column, + level shift
//           D  (row, 4 pixels): replication up-sampling, colour maths, 12/24-byte coalesced store
//   encode: 1  (row group, 4 pixels): coalesced sample load, RGB -> YCbCr, box mean, level shift
//           2  (block, row): forward transform of a row
//           3  (block, column): forward transform of a column, quantisation; the results go, as int16, to
//              their zig-zag places in the blocks' 128-byte lines, which lie over the float planes (written
//              once every thread has read its column)
//           4  (block, zig-zag octet): one 16-byte LDS read, one coalesced 16-byte store
// A thread's place inside its block (row / column / octet = tid & 7) is the same in every round of a stage
// (the rounds are 256 threads apart), so everything that depends on it -- zig-zag places, the block's LDS
// address -- is worked out once per thread.
// 8 threads share a block, so a wave holds 16 live values per lane instead of 128: occupancy is
// bounded by LDS (about 25 KB per workgroup), not by VGPRs.
#include <hip/hip_runtime.h>
#include <string.h>
#include "common.hpp"
#include "kernels.hpp"

typedef float f2 __attribute__((ext_vector_type(2)));

// frame_to_ycc at precision 8 in binary32 (see the file: why it equals the reference's binary64 expressions, and
// the two exhaustive checks), for one pixel and for two at a time (v_pk_mul_f32 / v_pk_fma_f32)
#include "ycc8.inc"
#define YCC8_SPLAT1(x) (x)
#define YCC8_SPLAT2(x) (f2{(x), (x)})
YCC8_DEFINE(__device__ __forceinline__, ycc8_one, float, YCC8_SPLAT1, __builtin_fmaf)
YCC8_DEFINE(__device__ __forceinline__, ycc8_two, f2, YCC8_SPLAT2, __builtin_elementwise_fma)

// zig-zag position -> raster index, usable in fully unrolled device code
__device__ static constexpr int kZig[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// the literal table as compile-time constants (so that the unrolled products use inline literals)
struct LutHolder {
	float v[8][8];
};

__device__ __forceinline__ constexpr LutHolder make_lut()
{
	return LutHolder{{
#include "dct_lut.inc"
	}};
}

// One 1-D transform of 8 values held in registers, two outputs at a time (v_pk_mul_f32 /
// v_pk_add_f32: unfused, each lane of the pair rounds like the scalar operation).
//   INV:  idct1 (imgproc.c:104-115)  out[x] = sum_u in[u] * lut[x][u]
//   !INV: fdct1 (imgproc.c:117-128)  out[u] = sum_x in[x] * lut[x][u]
template <bool INV>
__device__ __forceinline__ void dct1(const float (&in)[8], float (&out)[8])
{
	constexpr LutHolder LH = make_lut();
#pragma unroll
	for (int p = 0; p < 4; ++p) {
		const int o0 = 2 * p, o1 = 2 * p + 1;
		f2 acc = {0.f, 0.f};
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const f2 c = INV ? f2{LH.v[o0][i], LH.v[o1][i]} : f2{LH.v[i][o0], LH.v[i][o1]};
			const f2 prod = f2{in[i], in[i]} * c;
			acc = i == 0 ? prod : acc + prod;
		}
		out[o0] = acc.x;
		out[o1] = acc.y;
	}
}

#ifndef TILE_BLOCKS
#define TILE_BLOCKS 96
#endif
#ifndef TILE_PAD
#define TILE_PAD 4                  // floats behind every plane row: rows start on different banks (keeps 16-byte alignment)
#endif
#ifndef LINE_I16
#define LINE_I16 72                 // int16 from one block's parked zig-zag line to the next: 64 + 8, so that the lines of
                                    // the eight blocks a wave scatters coefficients into start on different banks
#endif
#define TILE_ROUNDS ((TILE_BLOCKS * 8 + 255) / 256)   // rounds of a (block, 1/8 block) stage over 256 threads

// (int)roundf(x) for |x| < 2^31: adding the largest float below one half (with the sign of x) and
// truncating gives round-half-away-from-zero for EVERY float (checked exhaustively on the host,
// tests/test_round_cpu.py); adding 0.5 itself would not (0.49999997 + 0.5 rounds to 1).
__device__ __forceinline__ int round_half_away(float x)
{
	return (int)(x + __builtin_copysignf(0x1.fffffep-2f, x));
}

// quantize_block() divides in binary32 (imgproc.c:28).  hipcc expands `n / d` into a reciprocal estimate, one
// Newton step on it, the quotient estimate and two correction steps (all FMAs), wrapped in scaling
// (v_div_scale) and a fix-up (v_div_fixup) for operands near the ends of the exponent range.  The first three
// steps depend on the divisor only and are done once per table entry (quant_reciprocal); with |n| < 2^24 and
// 1 <= d <= 65535 the scaling and fix-up steps are the identity unless the quotient is far below 0.5, where
// the rounding to an integer cannot see a difference, so quant_divide() gives the correctly rounded quotient
// wherever it matters.  jpeg_hip_selftest_divide() checks exactly that against the compiler's own `/` for
// every divisor (tests/test_gpu_stages.py).
__device__ __forceinline__ float quant_reciprocal(float d)
{
	const float r0 = __builtin_amdgcn_rcpf(d);
	const float e0 = __builtin_fmaf(-d, r0, 1.0f);
	return __builtin_fmaf(e0, r0, r0);
}

// nd = -d (the table holds it negated: the fused steps want -d, and a packed FMA would spend an instruction on
// the sign).  T: float, or f2 for two quotients at a time (v_pk_mul_f32 / v_pk_fma_f32: each half rounds like
// the scalar instruction).
template <typename T>
__device__ __forceinline__ T quant_divide(T n, T nd, T r1)
{
	const T q0 = n * r1;
	const T q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q0, n), r1, q0);
	return __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q1, n), r1, q1);
}

// ---- tile layout (host computed, shared by both kernels) -----------------------------------------
struct TileLayout {
	int G;                              // MCUs per workgroup
	unsigned b_magic16;                 // floor(2^16 / B) + 1: mcu = (block * b_magic16) >> 16 for block < 2^16 / B
	int TW, TH;                         // tile size in frame pixels
	unsigned qpr_magic;                 // floor(2^32 / (TW/4)) + 1
	int q_rows, q_rest;                 // 256 quads on = q_rows rows and q_rest quads on (a workgroup's stride in stage 1 / D)
	int base[JH_MAX_COMP], pitch[JH_MAX_COMP];          // component planes in LDS (floats)
	int koff[JH_MAX_MCU_BLOCKS];        // block k of MCU 0: offset of its top-left sample
	int kpitch[JH_MAX_MCU_BLOCKS];
	int kmst[JH_MAX_MCU_BLOCKS];        // 8 * H of the block's component: column step per MCU
	int ktq[JH_MAX_MCU_BLOCKS];         // quantisation table of the block's component
	int kta[JH_MAX_MCU_BLOCKS];         // its AC table id, 0 / 1 (encoder)
	int sx_magic[JH_MAX_COMP], sy_magic[JH_MAX_COMP];   // 65536 / step + 1: c = (p * magic) >> 16
	int lds_floats;
};

static void make_layout(const JhGeom &g, TileLayout *t)
{
	memset(t, 0, sizeof *t);
	int G = TILE_BLOCKS / g.B;
	G = G < 1 ? 1 : G;
	G = G > g.m_x ? g.m_x : G;
	t->G = G;
	t->b_magic16 = 65536u / (unsigned)g.B + 1u;
	t->TW = G * 8 * g.max_H;
	t->TH = 8 * g.max_V;
	t->qpr_magic = (unsigned)(0x100000000ull / (unsigned)(t->TW / 4)) + 1u;
	t->q_rows = 256 / (t->TW / 4);
	t->q_rest = 256 % (t->TW / 4);
	int off = 0;
	for (int c = 0; c < g.ncomp; ++c) {
		t->base[c] = off;
		t->pitch[c] = G * 8 * g.H[c] + TILE_PAD;
		off += 8 * g.V[c] * t->pitch[c];
		t->sx_magic[c] = 65536 / g.step_x[c] + 1;
		t->sy_magic[c] = 65536 / g.step_y[c] + 1;
	}
	t->lds_floats = off;
	for (int k = 0; k < g.B; ++k) {
		const int c = g.blk_comp[k];
		t->koff[k] = t->base[c] + g.blk_v[k] * 8 * t->pitch[c] + g.blk_h[k] * 8;
		t->kpitch[k] = t->pitch[c];
		t->kmst[k] = 8 * g.H[c];
		t->ktq[k] = g.Tq[c] & 3;
		t->kta[k] = g.Ta[c] & 1;
	}
}

struct BlockTask {
	int addr;      // LDS float index of the block's top-left sample
	int pitch;
	int tq;
};

// per-k tables staged in LDS (indexed per lane)
struct KTab {
	int koff[JH_MAX_MCU_BLOCKS], kpitch[JH_MAX_MCU_BLOCKS], kmst[JH_MAX_MCU_BLOCKS], ktq[JH_MAX_MCU_BLOCKS];
	int kta[JH_MAX_MCU_BLOCKS];         // AC table id (0 / 1) of the block's component
	uint32_t zig[8][2];                 // 8 raster indices (one byte each) per zig-zag octet
	uint8_t invz[64];                   // raster index -> zig-zag position
};

__device__ __forceinline__ void stage_ktab(KTab &kt, const TileLayout &L, int B, int tid)
{
	if (tid < B) {
		kt.koff[tid] = L.koff[tid];
		kt.kpitch[tid] = L.kpitch[tid];
		kt.kmst[tid] = L.kmst[tid];
		kt.ktq[tid] = L.ktq[tid];
		kt.kta[tid] = L.kta[tid];
	}
	if (tid >= 64 && tid < 72) {
		const int j = tid - 64;
		uint32_t lo = 0, hi = 0;
		for (int e = 0; e < 4; ++e) {
			lo |= (uint32_t)kZig[j * 8 + e] << (8 * e);
			hi |= (uint32_t)kZig[j * 8 + 4 + e] << (8 * e);
		}
		kt.zig[j][0] = lo;
		kt.zig[j][1] = hi;
	}
	if (tid >= 128 && tid < 192) {
		kt.invz[kZig[tid - 128]] = (uint8_t)(tid - 128);
	}
}

__device__ __forceinline__ BlockTask block_task(const KTab &kt, const TileLayout &L, int B, int b)
{
	// b < 128 blocks per tile, B <= 10: 24-bit multiplies (full rate) are exact
	const int mcu = B == 1 ? b : (int)(__umul24((unsigned)b, L.b_magic16) >> 16);
	const int k = b - (int)__umul24((unsigned)mcu, (unsigned)B);
	BlockTask r;
	r.pitch = kt.kpitch[k];
	r.addr = kt.koff[k] + (int)__umul24((unsigned)mcu, (unsigned)kt.kmst[k]);
	r.tq = kt.ktq[k];
	return r;
}

// ================================================================================== decode

struct DecPixArgs {
	const int16_t *coefs;      // [frame][nblk][64] zig-zag
	const JhQuant *quant;      // [frame], zig-zag order
	uint8_t *out;              // PPM/PGM bodies
	size_t frame_stride;
	float *tap;                // NULL, or [frame][Y][X][1 or 3]: the samples as floats before rounding (stage tests)
	TileLayout L;
	JhGeom g;
};

// 4 horizontally adjacent frame pixels of component c: transform_components_to_frame (frame.c:63-76)
// read backwards, frame(x, y) <- component(x / step_x, y / step_y)
// SX: the component's horizontal step when the launcher knows it (1, 2), 0 = look it up.  A known step
// lets the compiler see which of the four pixels share a chroma sample, and with it their colour products.
template <int SX>
__device__ __forceinline__ void fetch4(const float *tile, const TileLayout &L, const JhGeom &g, int c,
                                       int row, int px0, float (&s)[4])
{
	// 24-bit multiplies (row < 32, magic <= 65537, pitch < 2^12): full rate, the 32-bit one is not
	const unsigned cy = __umul24((unsigned)row, (unsigned)L.sy_magic[c]) >> 16;
	const float *p = tile + L.base[c] + __umul24(cy, (unsigned)L.pitch[c]);
	const int step = SX ? SX : g.step_x[c];
	if (step == 1) {
		const float4 v = *reinterpret_cast<const float4 *>(p + px0);
		s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
	} else if (step == 2) {
		const float2 v = *reinterpret_cast<const float2 *>(p + (px0 >> 1));
		s[0] = s[1] = v.x;
		s[2] = s[3] = v.y;
	} else if (step == 4) {
		s[0] = s[1] = s[2] = s[3] = p[px0 >> 2];
	} else {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			s[i] = p[__umul24((unsigned)(px0 + i), (unsigned)L.sx_magic[c]) >> 16];
		}
	}
}

// CSX: horizontal step of the components behind the first one when they all share it and the first has
// step 1 (what every subsampled YCbCr frame looks like), else 0
template <int NF, int CSX>
__global__ __launch_bounds__(256) void k_decode_pixels(const DecPixArgs a)
{
	extern __shared__ __align__(16) float tile[];
	__shared__ __align__(16) uint16_t sq[4][64];       // quantisation tables in RASTER order
	__shared__ KTab kt;

	const JhGeom &g = a.g;
	const TileLayout &L = a.L;
	const int tid = threadIdx.x;
	const int mx0 = blockIdx.x * L.G, my = blockIdx.y, f = blockIdx.z;
	const int B = g.B;
	const int nmcu_here = min(L.G, g.m_x - mx0);
	const int ntask = nmcu_here * B * 8;
	const int sub = tid & 7;                          // this thread's octet / row / column in every round

	for (int i = tid; i < 4 * 64; i += 256) {
		sq[i >> 6][kZig[i & 63]] = a.quant[f].q[i >> 6][i & 63];
	}
	stage_ktab(kt, L, B, tid);

	// ---- A: (block, zig-zag octet): the blocks' 128-byte lines as they are, parked over the planes ----
	{
		const int4 *src = reinterpret_cast<const int4 *>(
			a.coefs + ((size_t)f * g.nblk + (size_t)(my * g.m_x + mx0) * B) * 64);
		int4 *stg = reinterpret_cast<int4 *>(tile);
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				*reinterpret_cast<int4 *>(reinterpret_cast<int16_t *>(stg) + (t >> 3) * LINE_I16 + (t & 7) * 8) = src[t];
			}
		}
	}
	__syncthreads();

	BlockTask bts[TILE_ROUNDS];
#pragma unroll
	for (int it = 0; it < TILE_ROUNDS; ++it) {
		const int t = tid + it * 256;
		bts[it] = block_task(kt, L, B, (t < ntask ? t : 0) >> 3);
	}

	// ---- B: (block, row): dequantize_block (imgproc.c:17), rows first (imgproc.c:134-140) ---------
	{
		const int16_t *zp[8];                         // the zig-zag places of raster row `sub` in the line of this thread's
#pragma unroll                                        // block of round 0 (a round on: 32 lines on)
		for (int x = 0; x < 8; ++x) {
			zp[x] = reinterpret_cast<const int16_t *>(tile) + (tid >> 3) * LINE_I16 + kt.invz[sub * 8 + x];
		}
		float rowv[TILE_ROUNDS][8];
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				const uint4 qw = *reinterpret_cast<const uint4 *>(&sq[bts[it].tq][sub * 8]);
				const unsigned qs[4] = {qw.x, qw.y, qw.z, qw.w};
				float in[8];
#pragma unroll
				for (int x = 0; x < 8; ++x) {
					const int q = (int)((x & 1) ? (qs[x >> 1] >> 16) : (qs[x >> 1] & 0xffffu));
					// int32 product, then int -> float; |c| < 2^15 and q < 2^16: the 24-bit multiply is exact
					in[x] = (float)__mul24((int)zp[x][it * 32 * LINE_I16], q);
				}
				dct1<true>(in, rowv[it]);
			}
		}
		__syncthreads();                              // every coefficient has been read: the planes may be written
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				float4 *p = reinterpret_cast<float4 *>(tile + bts[it].addr + (int)__umul24((unsigned)sub, (unsigned)bts[it].pitch));
				p[0] = make_float4(rowv[it][0], rowv[it][1], rowv[it][2], rowv[it][3]);
				p[1] = make_float4(rowv[it][4], rowv[it][5], rowv[it][6], rowv[it][7]);
			}
		}
	}
	__syncthreads();

	// ---- C: (block, column): then columns (imgproc.c:142-148), level shift (imgproc.c:192-194) ---
	{
		const float shiftf = (float)(1 << (g.P - 1));
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				const unsigned pitch = (unsigned)bts[it].pitch;
				float *p = tile + bts[it].addr + sub;
				float in[8], o[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					in[u] = p[__umul24((unsigned)u, pitch)];
				}
				dct1<true>(in, o);
#pragma unroll
				for (int y = 0; y < 8; y += 2) {
					const f2 v = f2{o[y], o[y + 1]} + f2{shiftf, shiftf};
					p[__umul24((unsigned)y, pitch)] = v.x;
					p[__umul24((unsigned)y + 1u, pitch)] = v.y;
				}
			}
		}
	}
	__syncthreads();

	// ---- D: (row, 4 pixels): up-sampling, colour, rounding, store --------------------------------
	const int qpr = L.TW >> 2;                // quads per tile row
	const int X = g.X, Y = g.Y;
	constexpr int oc = NF == 1 ? 1 : 3;
	const float maxvalf = (float)((1 << g.P) - 1);
	const int shift = 1 << (g.P - 1);
	const int denom = 1 << g.P;
	(void)denom;
	(void)shift;
	// The tile's first sample in the output, and offsets from there by 24-bit multiplies (rowbytes < 2^19, row < 32)
	const unsigned pxbytes = (unsigned)(oc * g.sample_bytes), rowbytes = (unsigned)X * pxbytes;
	uint8_t *org = a.out + (size_t)f * a.frame_stride + (size_t)(my * L.TH) * rowbytes + (size_t)(mx0 * 8 * g.max_H) * pxbytes;

	// (row, quad) of this thread's first group by one division, of the next ones by stepping
	int row_n = (int)__umulhi((unsigned)tid, L.qpr_magic), quad_n = tid - row_n * qpr;
	for (int qd = tid; qd < L.TH * qpr; qd += 256) {
		const int row = row_n, px0 = quad_n * 4;
		quad_n += L.q_rest;
		row_n += L.q_rows + (quad_n >= qpr);
		quad_n -= quad_n >= qpr ? qpr : 0;
		const int gy = my * L.TH + row, gx0 = mx0 * 8 * g.max_H + px0;
		if (gy >= Y || gx0 >= X) {
			continue;
		}
		float s[NF][4];
#pragma unroll
		for (int c = 0; c < NF; ++c) {
			if (CSX != 0 && c == 0) {
				fetch4<1>(tile, L, g, c, row, px0, s[c]);
			} else if (CSX != 0 && (NF != 4 || c != NF - 1)) {
				fetch4<CSX>(tile, L, g, c, row, px0, s[c]);
			} else {
				fetch4<0>(tile, L, g, c, row, px0, s[c]);
			}
		}
		float ov[4 * oc];                         // the four pixels' samples before rounding
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			float o[oc];
			if constexpr (NF == 1) {
				o[0] = s[0][i];
			} else {
				// frame.c:204-206 / 226-228: (C - shift) is a float subtract, the rest is double
				const float Yl = s[0][i], Cb = s[1][i], Cr = s[2][i];
				const float r_ = Yl + 1.402 * (Cr - shift);
				const float g_ = Yl - 0.34414 * (Cb - shift) - 0.71414 * (Cr - shift);
				const float b_ = Yl + 1.772 * (Cb - shift);
				if constexpr (NF == 4) {
					const float K = s[NF - 1][i];
					o[0] = K - (r_ * K) / denom;       // frame.c:208-210 (all float)
					o[1] = K - (g_ * K) / denom;
					o[2] = K - (b_ * K) / denom;
				} else {
					o[0] = r_;
					o[1] = g_;
					o[2] = b_;
				}
			}
#pragma unroll
			for (int c = 0; c < oc; ++c) {
				ov[i * oc + c] = o[c];
			}
		}
		const int npx = min(4, X - gx0);
		if (a.tap) {
			// stage tap (tests only): frame->data behind frame_to_rgb (frame.c:188-244), the visible part, 1 or 3 channels
			float *tp = a.tap + (((size_t)f * Y + gy) * X + gx0) * oc;
#pragma unroll
			for (int i = 0; i < 4 * oc; ++i) {
				if (i < npx * oc) {
					tp[i] = ov[i];
				}
			}
		}
		uint8_t *p = org + (__umul24((unsigned)row, rowbytes) + __umul24((unsigned)px0, pxbytes));
		const bool whole = npx == 4 && ((reinterpret_cast<uintptr_t>(p) & 3) == 0);
		if (g.P == 8 && whole) {
			// (int)roundf(x) clamped to [0, 255] (frame.c:375-380) as v_cvt_pk_u8_f32 of floor(x + 0.49999997): the
			// instruction saturates at both ends and puts the byte where it belongs in the word (it rounds to
			// nearest by itself, hence the floor: a whole number converts as it is).  Below zero floor(x + c) <= 0
			// gives 0, as clamping first does; above 255 both give 255; in between it is round_half_away() on x >= 0.
			uint32_t *pw = reinterpret_cast<uint32_t *>(p);
#pragma unroll
			for (int wd = 0; wd < oc; ++wd) {
				unsigned wv = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					wv = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(ov[4 * wd + b] + 0x1.fffffep-2f), (unsigned)b, wv);
				}
				pw[wd] = wv;
			}
			continue;
		}
		int res[4 * oc];
#pragma unroll
		for (int n = 0; n < 4 * oc; ++n) {
			// (int)roundf(x) clamped to [0, maxval] (frame.c:375-380).  Rounding is monotonic and the bounds
			// are integers, so clamping first gives the same sample; then round_half_away() on x >= 0
			const float y = __builtin_amdgcn_fmed3f(ov[n], 0.f, maxvalf);
			res[n] = (int)(y + 0x1.fffffep-2f);
		}
		if (g.sample_bytes == 1) {
			if (whole) {
				uint32_t *pw = reinterpret_cast<uint32_t *>(p);
#pragma unroll
				for (int wd = 0; wd < oc; ++wd) {
					// three byte permutes (v_perm_b32: selector bytes 0-3 take from the second operand, 4-7 from the
					// first, 0x0c is zero); shifts and ors come out as five or six instructions
					const unsigned lo = __builtin_amdgcn_perm((unsigned)res[4 * wd + 1], (unsigned)res[4 * wd], 0x0c0c0400u);
					const unsigned hi = __builtin_amdgcn_perm((unsigned)res[4 * wd + 3], (unsigned)res[4 * wd + 2], 0x0c0c0400u);
					pw[wd] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
				}
			} else {
				// (unrolled: a run-time index would keep a second copy of res[] in registers the index can select from)
#pragma unroll
				for (int i = 0; i < 4 * oc; ++i) {
					if (i < npx * oc) {
						p[i] = (uint8_t)res[i];
					}
				}
			}
		} else {
			// big-endian 16-bit samples (htons, frame.c:386)
			if (whole) {
				uint32_t *pw = reinterpret_cast<uint32_t *>(p);
#pragma unroll
				for (int wd = 0; wd < 2 * oc; ++wd) {
					// bytes s0.1 s0.0 s1.1 s1.0 in one permute
					pw[wd] = __builtin_amdgcn_perm((unsigned)res[2 * wd + 1], (unsigned)res[2 * wd], 0x04050001u);
				}
			} else {
#pragma unroll
				for (int i = 0; i < 4 * oc; ++i) {
					if (i < npx * oc) {
						p[2 * i] = (uint8_t)(res[i] >> 8);
						p[2 * i + 1] = (uint8_t)res[i];
					}
				}
			}
		}
	}
}

int jh_launch_decode_pixels(const JhGeom &g, int nframes, const int16_t *coefs, const JhQuant *quant,
                            void *out, size_t frame_stride, hipStream_t stream, float *tap)
{
	DecPixArgs a;
	a.coefs = coefs;
	a.quant = quant;
	a.out = (uint8_t *)out;
	a.frame_stride = frame_stride;
	a.tap = tap;
	make_layout(g, &a.L);
	a.g = g;
	dim3 grid((g.m_x + a.L.G - 1) / a.L.G, g.m_y, nframes);
	const size_t lds = (size_t)a.L.lds_floats * sizeof(float);
	// chroma step known at compile time for the common layouts
	int csx = 0;
	if (g.Nf == 3 && g.step_x[0] == 1 && g.step_x[1] == g.step_x[2] && (g.step_x[1] == 1 || g.step_x[1] == 2)) {
		csx = g.step_x[1];
	}
	switch (g.Nf) {
	case 1:
		hipLaunchKernelGGL((k_decode_pixels<1, 0>), grid, dim3(256), lds, stream, a);
		break;
	case 3:
		if (csx == 1) {
			hipLaunchKernelGGL((k_decode_pixels<3, 1>), grid, dim3(256), lds, stream, a);
		} else if (csx == 2) {
			hipLaunchKernelGGL((k_decode_pixels<3, 2>), grid, dim3(256), lds, stream, a);
		} else {
			hipLaunchKernelGGL((k_decode_pixels<3, 0>), grid, dim3(256), lds, stream, a);
		}
		break;
	case 4:
		hipLaunchKernelGGL((k_decode_pixels<4, 0>), grid, dim3(256), lds, stream, a);
		break;
	default:
		return -1;
	}
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================== encode

struct EncPixArgs {
	const uint8_t *in;         // PPM/PGM bodies
	size_t frame_stride;
	const JhQuant *quant;      // [frame] RASTER order here (divisors are applied in raster order)
	int16_t *coefs;            // [frame][nblk][64] zig-zag
	unsigned long long *nzmap; // [frame][nblk] (with codes)
	uint32_t *acdc;            // [frame][nblk] (with codes)
	JhEncState *state;         // [frame], or NULL: a coefficient that does not fit 16 bits is reported there
	const JhRunTab *runtab;    // [frame] what AC symbols cost with the codes in force for the frame, or NULL
	float *tap;                // NULL, or [frame][component planes, by*8 rows of bx*8]: the planes behind the down-sampling,
	                           // before the level shift (stage tests)
	TileLayout L;
	JhGeom g;
};

// 4 horizontally adjacent samples-tuples of the input body as floats: read_frame_body (frame.c:295-345)
// incl. its padding rule (last column / last row replicated).  Fast path: whole dwords.
// `org` is the tile's first sample (inside the frame: tiles start at MCU corners), (tx, ty) the place inside the
// tile and (gx0, gy) = the tile's corner + (tx, ty) in the frame.  Offsets from `org` stay below 2^24 * 16
// and are made with 24-bit multiplies (rowbytes = X * NF * sb < 2^19, ty < 16): the full-width integer
// multiply runs at a quarter of the rate.
// whole: every one of the four lies inside the frame and the first sample's address is a multiple of 4
template <int NF>
__device__ __forceinline__ bool load4_whole(const uint8_t *org, unsigned rowbytes, int tx, int ty, int X, int Y, int sb, int gx0, int gy)
{
	const uint8_t *p = org + (__umul24((unsigned)ty, rowbytes) + __umul24((unsigned)tx, (unsigned)(NF * sb)));
	return gx0 + 3 < X && gy < Y && (reinterpret_cast<uintptr_t>(p) & 3) == 0;
}

template <int NF, int SB, bool WHOLE>
__device__ __forceinline__ void load4(const uint8_t *org, unsigned rowbytes, int tx, int ty, int X, int Y, int gx0, int gy,
                                      float (&s)[4][NF])
{
	if constexpr (WHOLE) {
		const uint8_t *p = org + (__umul24((unsigned)ty, rowbytes) + __umul24((unsigned)tx, (unsigned)(NF * SB)));
		const uint32_t *pw = reinterpret_cast<const uint32_t *>(p);
		if constexpr (SB == 1) {
			uint32_t w[NF];
#pragma unroll
			for (int i = 0; i < NF; ++i) {
				w[i] = pw[i];
			}
#pragma unroll
			for (int n = 0; n < 4 * NF; ++n) {
				s[n / NF][n % NF] = (float)((w[n >> 2] >> (8 * (n & 3))) & 0xff);
			}
		} else {
			uint32_t w[2 * NF];
#pragma unroll
			for (int i = 0; i < 2 * NF; ++i) {
				w[i] = pw[i];
			}
#pragma unroll
			for (int n = 0; n < 4 * NF; ++n) {
				const uint32_t h = (w[n >> 1] >> (16 * (n & 1))) & 0xffff;
				s[n / NF][n % NF] = (float)(((h & 0xff) << 8) | (h >> 8));   // ntohs (frame.c:321)
			}
		}
	} else {
		const int sy = min(gy, Y - 1) - (gy - ty);            // inside the tile again
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int sx = min(gx0 + i, X - 1) - (gx0 - tx);
			const uint8_t *p = org + (__umul24((unsigned)sy, rowbytes) + __umul24((unsigned)sx, (unsigned)(NF * SB)));
#pragma unroll
			for (int c = 0; c < NF; ++c) {
				s[i][c] = SB == 1 ? (float)p[c] : (float)((p[2 * c] << 8) | p[2 * c + 1]);
			}
		}
	}
}

// Stage 1 of the encode kernel for one group: SY rows of 4 pixels -> 4 luma values per row and 4 / SX values of each
// chroma plane, level-shifted, in the tile.  WHOLE: all of the group's samples lie inside the frame and its rows
// start on dword boundaries (every group of an aligned frame): dword loads, no padding rule, no per-pixel tests.
template <int NF, int SX, int SY, int SB, bool Y8, bool WHOLE>
__device__ __forceinline__ void encode_group(float *tile, const TileLayout &L, const uint8_t *org, unsigned rowbytes,
                                             int rg, int px0, int gx0, int my, int X, int Y, int shift, float shiftf,
                                             float *tapf, const JhGeom &g)
{
	float ch[2][SY][4];
#pragma unroll
	for (int sy = 0; sy < SY; ++sy) {
		const int row = rg * SY + sy, gy = my * L.TH + row;
		float s[4][NF];
		load4<NF, SB, WHOLE>(org, rowbytes, px0, row, X, Y, gx0, gy, s);
		float y4[4];
		if constexpr (NF == 3 && Y8) {
			// frame_to_ycc (frame.c:168-170) on exact integers and one correctly rounded division each
			// (ycc8.inc), two pixels per instruction.  Padding pixels keep their raw RGB values (quirk Q7).
			if (WHOLE) {
				// the pairs: pixels (0, 2) and (1, 3) where chroma is halved horizontally -- the box sums below then
				// add whole pairs -- else (0, 1) and (2, 3)
#pragma unroll
				for (int n = 0; n < 2; ++n) {
					const int i = SX == 2 ? n : 2 * n, j = SX == 2 ? n + 2 : 2 * n + 1;
					f2 yl, cb, cr;
					ycc8_two(f2{s[i][0], s[j][0]}, f2{s[i][1], s[j][1]}, f2{s[i][2], s[j][2]}, &yl, &cb, &cr);
					s[i][0] = yl.x; s[j][0] = yl.y;
					s[i][1] = cb.x; s[j][1] = cb.y;
					s[i][2] = cr.x; s[j][2] = cr.y;
				}
			} else {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					if (gx0 + i < X && gy < Y) {
						ycc8_one(s[i][0], s[i][1], s[i][2], &s[i][0], &s[i][1], &s[i][2]);
					}
				}
			}
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if constexpr (NF == 3) {
				if constexpr (!Y8) {
					if (WHOLE || (gx0 + i < X && gy < Y)) {
						// frame_to_ycc (frame.c:168-170): double arithmetic, left to right, rounded to
						// float.  Padding pixels keep their raw RGB values (loops stop at X, Y: quirk Q7).
						const float R = s[i][0], G_ = s[i][1], B_ = s[i][2];
						// The products with 0.5 are exact (a power of two), so `t + 0.5 * B` and `0.5 * R - t` round
						// once either way: written as fused operations they cost one instruction instead of two
						// and give the same binary64 value.  Every other product rounds on its own, as in C.
						const double Rd = R, Gd = G_, Bd = B_;
						const float Yl = 0.299 * Rd + 0.587 * Gd + 0.114 * Bd;
						const float Cb = __builtin_fma(0.5, Bd, - 0.1687 * Rd - 0.3313 * Gd) + shift;
						const float Cr = __builtin_fma(0.5, Rd, -(0.4187 * Gd)) - 0.0813 * Bd + shift;
						s[i][0] = Yl;
						s[i][1] = Cb;
						s[i][2] = Cr;
					}
				}
				ch[0][sy][i] = s[i][1];
				ch[1][sy][i] = s[i][2];
			}
			// luma: step 1 x 1: px = 0.f + s; px /= 1 (frame.c:114-123); then the level shift
			// of forward_dct (imgproc.c:220-222).  s is a sample or a luma value, never -0: 0.f + s == s
			y4[i] = s[i][0] - shiftf;
		}
		*reinterpret_cast<float4 *>(tile + L.base[0] + (int)__umul24((unsigned)row, (unsigned)L.pitch[0]) + px0) =
			make_float4(y4[0], y4[1], y4[2], y4[3]);
		if (tapf) {
			// stage tap (tests only): component[0].frame_buffer behind transform_frame_to_components (frame.c:84-132)
			float *tp = tapf + (size_t)gy * (g.bx[0] * 8) + gx0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				tp[i] = s[i][0];
			}
		}
	}
	if constexpr (NF == 3) {
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			float o[4 / SX];
#pragma unroll
			for (int ox = 0; ox < 4 / SX; ++ox) {
				// transform_frame_to_components (frame.c:114-123): px = 0.f; += (yy outer, xx inner); /= area.
				// The first term is a sample or `... + shift` in binary64, never -0: 0.f + it == it
				float px = ch[c][0][ox * SX];
#pragma unroll
				for (int yy = 0; yy < SY; ++yy) {
#pragma unroll
					for (int xx = 0; xx < SX; ++xx) {
						if (yy + xx) {
							px += ch[c][yy][ox * SX + xx];
						}
					}
				}
				px /= (float)(SY * SX);
				o[ox] = px - shiftf;
				if (tapf) {
					size_t base = (size_t)g.by[0] * 8 * g.bx[0] * 8;
					if (c == 1) {
						base += (size_t)g.by[1] * 8 * g.bx[1] * 8;
					}
					tapf[base + (size_t)(my * (L.TH / SY) + rg) * (g.bx[c + 1] * 8) + gx0 / SX + ox] = px;
				}
			}
			float *dst = tile + L.base[c + 1] + (int)__umul24((unsigned)rg, (unsigned)L.pitch[c + 1]) + px0 / SX;
			if constexpr (SX == 1) {
				*reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
			} else {
				*reinterpret_cast<float2 *>(dst) = make_float2(o[0], o[1]);
			}
		}
	}
}

// bits 0 ... 7 (and rubbish above them): which of the eight int16 of v are not zero.  min(x, 1) on both halves
// of a dword leaves the flags in bits 0 and 16; the four dwords go two bits apart, the high halves come down by
// 15.  (Written as an instruction: the compiler turns min(x, 1) into compare + select + permute, three times
// the work.)
__device__ __forceinline__ unsigned nz_octet(const int4 v)
{
	const unsigned ones = 0x00010001u;
	unsigned a, b, c, d;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(a) : "v"(v.x), "v"(ones));
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(b) : "v"(v.y), "v"(ones));
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(c) : "v"(v.z), "v"(ones));
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(v.w), "v"(ones));
	const unsigned acc = a | (b << 2) | (c << 4) | (d << 6);
	return acc | (acc >> 15);
}

// Stage 5 of the encode kernel (bits of every block's AC symbols) works in the part of the tile's LDS that
// lies behind the blocks' int16 lines and is free once the float planes are dead:
//   nzm    the blocks' nonzero maps, one byte per octet
//   pbits  AC bits of the two halves of a block
//   acfix  sizes of EOB / ZRL per table id
//   runtab what a coefficient of category c behind a zero run of r = 0 ... 63 costs: size + category of the symbol
//          (r & 15, c) plus (r >> 4) times the size of ZRL -- one look-up per nonzero coefficient, nothing to mask
//          or multiply.  A copy of the frame's JhRunTab (k_enc_runtab makes it, once per run(): built here, by every
//          workgroup from the codes, the two tables cost two barriers and 0.2 of the kernel's 2.8 ms).
struct Stage5 {
	unsigned long long nzm[128];
	unsigned pbits[2][128];
	uint16_t acfix[2][2];
	uint16_t pad[4];
	__attribute__((aligned(16))) uint16_t runtab[2][64][16];
};
#define STAGE5_OFF (TILE_BLOCKS * LINE_I16 * 2)             // bytes: behind the lines of a full tile

// JhRunTab of every frame from its AC codes (codes[frame * 4 + 2 + id]): one workgroup per frame
__global__ __launch_bounds__(256) void k_enc_runtab(const JhHuffEnc *codes, JhRunTab *out)
{
	__shared__ __attribute__((aligned(16))) uint16_t actab[2][256];    // size + category per symbol (0: never looked up)
	__shared__ uint16_t acfix[2][2];
	const int f = blockIdx.x, tid = threadIdx.x;
	for (int i = tid; i < 2 * 256; i += 256) {
		const int th = i >> 8, sym = i & 255, cat = sym & 15;
		const unsigned size = codes[(size_t)f * 4 + 2 + th].code[sym] & 0xffu;
		actab[th][sym] = (uint16_t)(cat == 0 ? 0u : (size ? size + (unsigned)cat : 0x4000u));
		if (sym == 0x00) {
			acfix[th][0] = (uint16_t)(size ? size : 0x4000u);
		}
		if (sym == 0xf0) {
			acfix[th][1] = (uint16_t)(size ? size : 0x4000u);
		}
	}
	__syncthreads();
	// thread = (table, run, eight categories), one 16-byte read, one 16-byte write.  A flagged entry (0x4000) stays
	// flagged: at most 3 * 16 is added.
	const int th = tid >> 7, run = (tid >> 1) & 63, half = tid & 1;
	const unsigned z = (unsigned)(run >> 4) * acfix[th][1], zz = z | (z << 16);
	uint4 v = *reinterpret_cast<const uint4 *>(&actab[th][((run & 15) << 4) + half * 8]);
	v.x += zz; v.y += zz; v.z += zz; v.w += zz;            // no carry between the halves: every entry < 0x8000
	*reinterpret_cast<uint4 *>(&out[f].run[th][run][half * 8]) = v;
	if (tid < 4) {
		(&out[f].fix[0][0])[tid] = (&acfix[0][0])[tid];
	}
}

// NF = 1: one plane, no sub-sampling.  NF = 3: luma at full resolution, both chroma planes
// down-sampled SX x SY (what read_image() sets up: luma H x V, chroma 1 x 1; encoder.c:109-146).
// Y8: 8-bit samples at precision 8 -- frame_to_ycc without binary64 (ycc8.inc).
// SB: bytes per sample
template <int NF, int SX, int SY, int SB, bool Y8>
__global__ __launch_bounds__(256) void k_encode_pixels(const EncPixArgs a)
{
	extern __shared__ __align__(16) float tile[];
	// Per table (the encoder's frames use tables 0 / 1 only, encoder.c:133-135), raster column c and pair of rows
	// (2p, 2p + 1): the two divisors, negated, and their refined reciprocals (see quant_divide) -- what a thread
	// of stage 3 wants for two of its coefficients, in one 16-byte read.
	__shared__ float4 sqf[2][8][4];
	__shared__ KTab kt;
	Stage5 &s5 = *reinterpret_cast<Stage5 *>(reinterpret_cast<char *>(tile) + STAGE5_OFF);

	const JhGeom &g = a.g;
	const TileLayout &L = a.L;
	const int tid = threadIdx.x;
	const int mx0 = blockIdx.x * L.G, my = blockIdx.y, f = blockIdx.z;
	const int B = g.B;
	const int nmcu_here = min(L.G, g.m_x - mx0);
	const int ntask = nmcu_here * B * 8;
	const int X = g.X, Y = g.Y;
	const int shift = 1 << (g.P - 1);
	const float shiftf = (float)shift;
	const unsigned rowbytes = (unsigned)X * (unsigned)(NF * SB);
	// the tile's first sample
	const uint8_t *org = a.in + (size_t)f * a.frame_stride + (size_t)(my * L.TH) * rowbytes +
		(size_t)(mx0 * 8 * g.max_H) * (size_t)(NF * SB);

	for (int i = tid; i < 2 * 64; i += 256) {
		const float d = (float)a.quant[f].q[i >> 6][i & 63];
		const int u = (i >> 3) & 7, c = i & 7;
		float *e = reinterpret_cast<float *>(&sqf[i >> 6][c][u >> 1]) + (u & 1);
		e[0] = -d;
		e[2] = quant_reciprocal(d);
	}
	stage_ktab(kt, L, B, tid);
	uint4 rt = make_uint4(0, 0, 0, 0);                // stage 5's tables, 16 bytes per thread
	unsigned rfix = 0;

	// ---- 1: (row group, 4 pixels): samples -> float, RGB -> YCbCr, box mean, level shift --------
	{
		size_t plane_floats = 0;
		for (int c = 0; c < NF; ++c) {
			plane_floats += (size_t)g.by[c] * 8 * g.bx[c] * 8;
		}
		float *tapf = a.tap ? a.tap + (size_t)f * plane_floats : nullptr;
		const int qpr = L.TW >> 2;
		const int wpx = nmcu_here * 8 * g.max_H;
		const int ngroups = (L.TH / SY) * qpr;
		int rg_n = (int)__umulhi((unsigned)tid, L.qpr_magic), quad_n = tid - rg_n * qpr;
		for (int qd = tid; qd < ngroups; qd += 256) {
			const int rg = rg_n, px0 = quad_n * 4;
			quad_n += L.q_rest;
			rg_n += L.q_rows + (quad_n >= qpr);
			quad_n -= quad_n >= qpr ? qpr : 0;
			if (px0 >= wpx) {
				continue;
			}
			const int gx0 = mx0 * 8 * g.max_H + px0;
			bool whole = true;
#pragma unroll
			for (int sy = 0; sy < SY; ++sy) {
				whole = whole && load4_whole<NF>(org, rowbytes, px0, rg * SY + sy, X, Y, SB, gx0, my * L.TH + rg * SY + sy);
			}
			if (whole) {
				encode_group<NF, SX, SY, SB, Y8, true>(tile, L, org, rowbytes, rg, px0, gx0, my, X, Y, shift, shiftf, tapf, g);
			} else {
				encode_group<NF, SX, SY, SB, Y8, false>(tile, L, org, rowbytes, rg, px0, gx0, my, X, Y, shift, shiftf, tapf, g);
			}
		}
	}
	__syncthreads();

	BlockTask bts[TILE_ROUNDS];
	const int sub = tid & 7;                          // this thread's row / column / octet in every round
#pragma unroll
	for (int it = 0; it < TILE_ROUNDS; ++it) {
		const int t = tid + it * 256;
		bts[it] = block_task(kt, L, B, (t < ntask ? t : 0) >> 3);
	}

	// ---- 2: (block, row): rows first (imgproc.c:155-161) -----------------------------------------
#pragma unroll
	for (int it = 0; it < TILE_ROUNDS; ++it) {
		const int t = tid + it * 256;
		if (t < ntask) {
			float4 *p = reinterpret_cast<float4 *>(tile + bts[it].addr + (int)__umul24((unsigned)sub, (unsigned)bts[it].pitch));
			const float4 v0 = p[0], v1 = p[1];
			const float in[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
			float o[8];
			dct1<false>(in, o);
			p[0] = make_float4(o[0], o[1], o[2], o[3]);
			p[1] = make_float4(o[4], o[5], o[6], o[7]);
		}
	}
	__syncthreads();

	// ---- 3: (block, column): columns (imgproc.c:163-169), then quantize_block (imgproc.c:28) -------
	{
		int zcol[8];                                  // zig-zag places of raster column `sub`
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			zcol[u] = kt.invz[u * 8 + sub];
		}
		int iq[TILE_ROUNDS][8];
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				const unsigned pitch = (unsigned)bts[it].pitch;
				const float *p = tile + bts[it].addr + sub;
				const float4 *q = sqf[bts[it].tq & 1][sub];
				float in[8], o[8];
#pragma unroll
				for (int y = 0; y < 8; ++y) {
					in[y] = p[__umul24((unsigned)y, pitch)];
				}
				dct1<false>(in, o);
#pragma unroll
				for (int u = 0; u < 8; u += 2) {
					// IEEE divide, round half away from zero (imgproc.c:28)
					const float4 dr = q[u >> 1];
					const f2 qq = quant_divide(f2{o[u], o[u + 1]}, f2{dr.x, dr.y}, f2{dr.z, dr.w});
					// round_half_away() on the pair
					const f2 rr = qq + f2{__builtin_copysignf(0x1.fffffep-2f, qq.x), __builtin_copysignf(0x1.fffffep-2f, qq.y)};
					iq[it][u] = (int)rr.x;
					iq[it][u + 1] = (int)rr.y;
				}
			}
		}
		// Coefficients are stored in 16 bits.  Up to 12 bits of sample precision nothing larger can come out of the
		// DCT; beyond (P 13 ... 16 with small divisors) it can: category 16 and up, for which no Huffman table has
		// a code (coeffs.c:56-74, huffman.c:252) -- the frame fails with that error.
		if (a.state) {                                // P > 12 only
			int wide = 0;                             // negative <=> some |coefficient| > 32767
#pragma unroll
			for (int it = 0; it < TILE_ROUNDS; ++it) {
				if (tid + it * 256 < ntask) {
#pragma unroll
					for (int u = 0; u < 8; ++u) {
						wide |= (iq[it][u] + 32767) | (32767 - iq[it][u]);
					}
				}
			}
			if (wide < 0) {
				atomicOr(&a.state[f].status, (uint32_t)JH_EST_NO_CODE);
			}
		}
		// stage 5's tables: asked for here -- the transforms, which need every register, are done -- and in hand
		// two barriers on
		if (a.runtab) {
			rt = reinterpret_cast<const uint4 *>(&a.runtab[f].run[0][0][0])[tid];
			rfix = (&a.runtab[f].fix[0][0])[tid & 3];
		}
		__syncthreads();                              // every column has been read: the blocks' int16 lines go over the planes
		uint16_t *zp[8];                              // this thread's places in the line of its block of round 0
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			zp[u] = reinterpret_cast<uint16_t *>(tile) + (tid >> 3) * LINE_I16 + zcol[u];
		}
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			if (tid + it * 256 < ntask) {
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					zp[u][it * 32 * LINE_I16] = (uint16_t)iq[it][u];    // block (tid + it * 256) >> 3
				}
			}
		}
	}
	__syncthreads();

	// ---- 4: (block, zig-zag octet): the finished lines, 16 bytes per thread, coalesced ------------
	{
		int4 *dst = reinterpret_cast<int4 *>(
			a.coefs + ((size_t)f * g.nblk + (size_t)(my * g.m_x + mx0) * B) * 64);
		const int4 *stg = reinterpret_cast<const int4 *>(tile);
#pragma unroll
		for (int it = 0; it < TILE_ROUNDS; ++it) {
			const int t = tid + it * 256;
			if (t < ntask) {
				const int4 v = *reinterpret_cast<const int4 *>(reinterpret_cast<const int16_t *>(stg) + (t >> 3) * LINE_I16 + (t & 7) * 8);
				dst[t] = v;
				if (a.runtab) {
					reinterpret_cast<uint8_t *>(s5.nzm)[t] = (uint8_t)nz_octet(v);
				}
			}
		}
	}
	if (!a.runtab) {
		return;
	}
#ifdef X_SKIP5
	return;
#endif
	reinterpret_cast<uint4 *>(&s5.runtab[0][0][0])[tid] = rt;  // 256 threads x 16 bytes = the two tables
	if (tid < 4) {
		(&s5.acfix[0][0])[tid] = (uint16_t)rfix;
	}
	__syncthreads();

	// ---- 5: (block, half): what write_ac() will spend on the block (coeffs.c:164-194, Figure F.2), walked over
	// the nonzero coefficients only (the zero run is the distance to the previous one).  Two lanes per block,
	// coefficients 1..31 and 32..63, the halves in different lanes (the first half holds most of the work);
	// the next coefficient is fetched while the current symbol is priced.
	static_assert(TILE_BLOCKS <= 128, "stage 5 maps a tile's blocks to 128 lanes");
	const int nb = ntask >> 3;
	{
		// first halves on threads 0 .. nb - 1, second halves (few nonzeros) behind them: three waves instead of two
		// pairs of a full and a half-empty one, and the fourth is done
		const int h = tid >= nb ? 1 : 0, b = h ? tid - nb : tid;
		if (b < nb) {
			const int mcu = B == 1 ? b : (int)(__umul24((unsigned)b, L.b_magic16) >> 16);
			const int k = b - (int)__umul24((unsigned)mcu, (unsigned)B);
			const int ta = kt.kta[k];
			const uint16_t *T = &s5.runtab[ta][0][0];
			const unsigned eob = s5.acfix[ta][0];
			const int16_t *line = reinterpret_cast<const int16_t *>(tile) + b * LINE_I16 + h * 32;
			const unsigned long long map = s5.nzm[b];
			const unsigned lo = (unsigned)map & ~1u, hi = (unsigned)(map >> 32);
			unsigned m = h ? hi : lo;
			// position (inside this half) of the nonzero coefficient before the first of this half; DC counts as one
			int prev = h ? (31 - __builtin_clz(lo | 1u)) - 32 : 0;
			unsigned bits = 0, tprev = 0;
			// (bit 31 ORed in: an empty map reads coefficient 31 for nothing instead of taking a compare and a select)
			int z = __builtin_ctz(m | 0x80000000u);
			int c = line[z];
			while (m) {
				m &= m - 1;
				const int zn = __builtin_ctz(m | 0x80000000u);
				const int cn = line[zn];
				const int run = z - prev - 1;
				prev = z;
				const int cat = __builtin_amdgcn_frexp_expf((float)c);  // bits of |c| (encode_cat, coeffs.c:56-74)
				bits += tprev;
				tprev = T[(run << 4) + cat];
				z = zn;
				c = cn;
			}
			bits += tprev;
			if (h && !(hi >> 31)) {
				bits += eob;
			}
			s5.pbits[h][b] = bits;
		}
	}
	__syncthreads();
	if (tid < nb) {
		const unsigned bits = s5.pbits[0][tid] + s5.pbits[1][tid];
		const size_t blk = (size_t)f * g.nblk + (size_t)(my * g.m_x + mx0) * B + tid;
		const unsigned dc = (unsigned)(uint16_t)reinterpret_cast<const int16_t *>(tile)[tid * LINE_I16];
		a.nzmap[blk] = s5.nzm[tid];
		a.acdc[blk] = (bits >= 0x4000u ? JH_ACBITS_NOCODE : bits) | (dc << 16);
	}
}

int jh_launch_encode_pixels(const JhGeom &g, int nframes, const void *in, size_t frame_stride,
                            const JhQuant *quant_raster, int16_t *coefs, unsigned long long *nzmap, uint32_t *acdc,
                            const JhHuffEnc *codes, JhRunTab *runtab, JhEncState *state, hipStream_t stream, float *tap)
{
	EncPixArgs a;
	a.tap = tap;
	a.in = (const uint8_t *)in;
	a.frame_stride = frame_stride;
	a.quant = quant_raster;
	a.coefs = coefs;
	a.nzmap = nzmap;
	a.acdc = acdc;
	a.runtab = nzmap && acdc && codes ? runtab : nullptr;
	if (a.runtab) {
		hipLaunchKernelGGL(k_enc_runtab, dim3(nframes), dim3(256), 0, stream, codes, runtab);
	}
	a.state = g.P > 12 ? state : nullptr;
	make_layout(g, &a.L);
	a.g = g;
	dim3 grid((g.m_x + a.L.G - 1) / a.L.G, g.m_y, nframes);
	size_t lds = (size_t)a.L.lds_floats * sizeof(float);
	if (a.runtab && lds < STAGE5_OFF + sizeof(Stage5)) {
		lds = STAGE5_OFF + sizeof(Stage5);                 // small tiles: room for stage 5 behind the lines
	}
	if (g.sample_bytes != 1 && g.sample_bytes != 2) {
		return -1;
	}
	// 0: 16-bit samples, 1: 8-bit samples, 2: 8-bit samples at precision 8 (frame_to_ycc in binary32: ycc8.inc)
	const int mode = g.sample_bytes == 2 ? 0 : (g.P == 8 && g.Nf == 3 ? 2 : 1);
#define JH_ENC_LAUNCH(NF, SX, SY)                                                                              \
	do {                                                                                                       \
		if (mode == 0) {                                                                                       \
			hipLaunchKernelGGL((k_encode_pixels<NF, SX, SY, 2, false>), grid, dim3(256), lds, stream, a);      \
		} else if (mode == 1) {                                                                                \
			hipLaunchKernelGGL((k_encode_pixels<NF, SX, SY, 1, false>), grid, dim3(256), lds, stream, a);      \
		} else {                                                                                               \
			hipLaunchKernelGGL((k_encode_pixels<NF, SX, SY, 1, NF == 3>), grid, dim3(256), lds, stream, a);    \
		}                                                                                                      \
	} while (0)
	if (g.Nf == 1) {
		JH_ENC_LAUNCH(1, 1, 1);
	} else if (g.Nf == 3 && g.step_x[0] == 1 && g.step_y[0] == 1 && g.step_x[1] == g.max_H && g.step_y[1] == g.max_V &&
	           g.step_x[2] == g.max_H && g.step_y[2] == g.max_V && g.max_H <= 2 && g.max_V <= 2) {
		const int code = g.max_H * 10 + g.max_V;
		switch (code) {
		case 11:
			JH_ENC_LAUNCH(3, 1, 1);
			break;
		case 21:
			JH_ENC_LAUNCH(3, 2, 1);
			break;
		case 12:
			JH_ENC_LAUNCH(3, 1, 2);
			break;
		default:
			JH_ENC_LAUNCH(3, 2, 2);
			break;
		}
	} else {
		return -1;   // jpeg_hip_encoder_configure() admits nothing else (encoder.c:123-124)
	}
#undef JH_ENC_LAUNCH
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================== self-test

// quant_divide() and round_half_away() against the compiler's own division and roundf(), for EVERY divisor
// 1 .. 65535: numerators around every rounding tie (k + 0.5) * d that fits 2^24, k stepping so that about
// `per` ties per divisor are visited, each with its float neighbours, plus pseudo-random numerators.
// out[0] = cases, out[1] = quotients that differ while not both far below 0.5, out[2] = rounded results that differ
__global__ __launch_bounds__(256) void k_selftest_divide(unsigned long long *out, int per, unsigned seed)
{
	const int d_i = blockIdx.x * 256 + threadIdx.x + 1;
	if (d_i > 65535) {
		return;
	}
	const float d = (float)d_i, r1 = quant_reciprocal(d);
	unsigned long long cases = 0, badq = 0, badr = 0;
	const int kmax = (int)(16777216.0f / d);
	const int step = kmax / per + 1;
	unsigned h = seed ^ (unsigned)d_i * 0x9E3779B1u;
	for (int k = 0; k <= kmax; k += step) {
		const float tie = ((float)k + 0.5f) * d;
		for (int nb = -2; nb <= 2; ++nb) {
			for (int sgn = 0; sgn < 2; ++sgn) {
				float n = __int_as_float(__float_as_int(tie) + nb);
				n = sgn ? -n : n;
				const float want = n / d, got = quant_divide(n, -d, r1);
				cases++;
				if (want != got && !(fabsf(want) < 0.25f && fabsf(got) < 0.25f)) {
					badq++;
				}
				if ((int)roundf(want) != round_half_away(got)) {
					badr++;
				}
			}
		}
		h = h * 1664525u + 1013904223u;
		const float n = (float)((int)(h >> 7) - (1 << 24)) * 0.5f;     // |n| < 2^24, halves included
		const float want = n / d, got = quant_divide(n, -d, r1);
		cases++;
		if (want != got && !(fabsf(want) < 0.25f && fabsf(got) < 0.25f)) {
			badq++;
		}
		if ((int)roundf(want) != round_half_away(got)) {
			badr++;
		}
	}
	atomicAdd(&out[0], cases);
	atomicAdd(&out[1], badq);
	atomicAdd(&out[2], badr);
}

// ycc8_one() / ycc8_two() against the binary64 expressions of frame_to_ycc (frame.c:168-170) as the compiler
// evaluates them (unfused: -ffp-contract=off), for all 2^24 (R, G, B): thread = (R, G), loop over B in pairs.
// out[0] = triples, out[1] = triples where the one-pixel form differs in any of the three bit patterns,
// out[2] = the same for the two-pixel form
__global__ __launch_bounds__(256) void k_selftest_ycc8(unsigned long long *out)
{
	const int rg = blockIdx.x * 256 + threadIdx.x;
	const float R = (float)(rg >> 8), G = (float)(rg & 255);
	const int shift = 128;
	unsigned long long bad1 = 0, bad2 = 0;
	for (int b = 0; b < 256; b += 2) {
		float w[2][3];
		for (int i = 0; i < 2; ++i) {
			const float B = (float)(b + i);
			w[i][0] = 0.299 * R + 0.587 * G + 0.114 * B;
			w[i][1] = - 0.1687 * R - 0.3313 * G + 0.5 * B + shift;
			w[i][2] = 0.5 * R - 0.4187 * G - 0.0813 * B + shift;
			float y, cb, cr;
			ycc8_one(R, G, B, &y, &cb, &cr);
			bad1 += __float_as_uint(y) != __float_as_uint(w[i][0]) || __float_as_uint(cb) != __float_as_uint(w[i][1]) ||
			        __float_as_uint(cr) != __float_as_uint(w[i][2]);
		}
		f2 y, cb, cr;
		ycc8_two(f2{R, R}, f2{G, G}, f2{(float)b, (float)(b + 1)}, &y, &cb, &cr);
		bad2 += __float_as_uint(y.x) != __float_as_uint(w[0][0]) || __float_as_uint(cb.x) != __float_as_uint(w[0][1]) ||
		        __float_as_uint(cr.x) != __float_as_uint(w[0][2]);
		bad2 += __float_as_uint(y.y) != __float_as_uint(w[1][0]) || __float_as_uint(cb.y) != __float_as_uint(w[1][1]) ||
		        __float_as_uint(cr.y) != __float_as_uint(w[1][2]);
	}
	atomicAdd(&out[0], 256ull);
	atomicAdd(&out[1], bad1);
	atomicAdd(&out[2], bad2);
}

int jh_launch_selftest_ycc8(unsigned long long *d_out, hipStream_t stream)
{
	hipLaunchKernelGGL(k_selftest_ycc8, dim3(256), dim3(256), 0, stream, d_out);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

int jh_launch_selftest_divide(unsigned long long *d_out, int per, unsigned seed, hipStream_t stream)
{
	hipLaunchKernelGGL(k_selftest_divide, dim3(256), dim3(256), 0, stream, d_out, per, seed);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ================================================================================== the stages one by one
//
// dequantize / inverse_dct / forward_dct / quantize (imgproc.h:8-16) and frame_to_rgb / frame_to_ycc (frame.h:21,33) as
// kernels of their own, on the reference's own buffer layouts (flt_block / int_block arrays, the interleaved float
// frame): what integration/ref_glue.c (-DGLUE_STAGES) puts behind the reference's function names, so that every FLOAT
// stage boundary of the reference exists on the GPU as well.  Parity vehicles: the fused kernels above are the fast
// path.  Same arithmetic as there: dct1<>, the literal table, no contraction, binary64 where C promotes.

// dequantize_block (imgproc.c:10-19): f[j] = (float)(c[j] * (int32)Q[j]), Q in raster order
__global__ __launch_bounds__(256) void k_stage_dequantize(const int32_t *coef, const uint16_t *Q, float *out, size_t n)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		out[i] = (float)(int32_t)((uint32_t)coef[i] * (uint32_t)Q[i & 63]);      // (wraps like the int32 product does)
	}
}

// quantize_block (imgproc.c:21-30): c[j] = (int32)roundf(f[j] / (float)Q[j]) -- the compiler's correctly rounded
// division and round_half_away() (== roundf for every float, tests/test_round_cpu.py)
__global__ __launch_bounds__(256) void k_stage_quantize(const float *in, const uint16_t *Q, int32_t *out, size_t n)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		out[i] = round_half_away(in[i] / (float)Q[i & 63]);
	}
}

// inverse_dct (imgproc.c:172-200): idct (rows, then columns, :130-149), then c[j] += shift;
// forward_dct (imgproc.c:202-230): c[j] -= shift, then fdct (rows, then columns, :151-170).  One thread per block.
template <bool INV>
__global__ __launch_bounds__(64) void k_stage_dct(float *blocks, size_t nblocks, float shiftf)
{
	const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
	if (b >= nblocks) {
		return;
	}
	float *p = blocks + b * 64;
	float c[8][8];
#pragma unroll
	for (int v = 0; v < 8; ++v) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			c[v][u] = INV ? p[v * 8 + u] : p[v * 8 + u] - shiftf;
		}
	}
#pragma unroll
	for (int v = 0; v < 8; ++v) {                             // rows
		float o[8];
		dct1<INV>(c[v], o);
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			c[v][u] = o[u];
		}
	}
#pragma unroll
	for (int u = 0; u < 8; ++u) {                             // columns
		float in[8], o[8];
#pragma unroll
		for (int v = 0; v < 8; ++v) {
			in[v] = c[v][u];
		}
		dct1<INV>(in, o);
#pragma unroll
		for (int v = 0; v < 8; ++v) {
			p[v * 8 + u] = INV ? o[v] + shiftf : o[v];
		}
	}
}

// frame_to_rgb (frame.c:188-244) / frame_to_ycc (frame.c:154-186) on the interleaved float frame, x < X and y < Y only
template <bool TO_RGB>
__global__ __launch_bounds__(256) void k_stage_colour(float *data, int comps, int X, int Y, size_t size_x, int P)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)X * Y) {
		return;
	}
	const size_t y = i / (size_t)X, x = i - y * (size_t)X;
	float *px = data + (y * size_x + x) * comps;
	const int shift = 1 << (P - 1), denom = 1 << P;
	if (TO_RGB) {
		const float Yl = px[0], Cb = px[1], Cr = px[2];
		const float r_ = Yl + 1.402 * (Cr - shift);
		const float g_ = Yl - 0.34414 * (Cb - shift) - 0.71414 * (Cr - shift);
		const float b_ = Yl + 1.772 * (Cb - shift);
		if (comps == 4) {
			const float K = px[3];
			px[0] = K - (r_ * K) / denom;
			px[1] = K - (g_ * K) / denom;
			px[2] = K - (b_ * K) / denom;
			px[3] = 255.0f;
		} else {
			px[0] = r_;
			px[1] = g_;
			px[2] = b_;
		}
	} else {
		const float R = px[0], G_ = px[1], B_ = px[2];
		const float Yl = 0.299 * R + 0.587 * G_ + 0.114 * B_;
		const float Cb = - 0.1687 * R - 0.3313 * G_ + 0.5 * B_ + shift;
		const float Cr = 0.5 * R - 0.4187 * G_ - 0.0813 * B_ + shift;
		px[0] = Yl;
		px[1] = Cb;
		px[2] = Cr;
	}
}

int jh_launch_stage(int what, void *a, const void *b, void *c, size_t n, int p0, int p1, int p2, size_t p3, hipStream_t s)
{
	switch (what) {
	case JH_STAGE_DEQUANTIZE:
		hipLaunchKernelGGL(k_stage_dequantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int32_t *)a, (const uint16_t *)b, (float *)c, n);
		break;
	case JH_STAGE_QUANTIZE:
		hipLaunchKernelGGL(k_stage_quantize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)a, (const uint16_t *)b, (int32_t *)c, n);
		break;
	case JH_STAGE_IDCT:
		hipLaunchKernelGGL(k_stage_dct<true>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (float *)a, n, (float)(1 << (p0 - 1)));
		break;
	case JH_STAGE_FDCT:
		hipLaunchKernelGGL(k_stage_dct<false>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, (float *)a, n, (float)(1 << (p0 - 1)));
		break;
	case JH_STAGE_TO_RGB:
		hipLaunchKernelGGL(k_stage_colour<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float *)a, p0, p1, p2, p3, (int)(size_t)b);
		break;
	case JH_STAGE_TO_YCC:
		hipLaunchKernelGGL(k_stage_colour<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float *)a, p0, p1, p2, p3, (int)(size_t)b);
		break;
	default:
		return -1;
	}
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
