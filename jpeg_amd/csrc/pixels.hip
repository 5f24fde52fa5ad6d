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
// uses the literal table T1 and the reference's summation order (s = 0.f; s += in*lut, index
// ascending), colour maths runs in binary64 exactly where the reference's C expressions promote
// to double, and fp32 division is hipcc's correctly rounded default.  MFMA is deliberately not
// used: its fused accumulation cannot reproduce the reference's roundings.
//
// Work decomposition (one workgroup = a run of G MCUs of one MCU row, 256-pixel wide tiles):
//   phase 1  one thread per 8x8 block: the whole 2-D transform lives in VGPRs (64 values, all
//            indices compile-time), coefficients come straight from HBM as 8 x 16-byte loads of the
//            block's own 128-byte line.
//   phase 2  one thread per 4 horizontally adjacent pixels: colour maths from the LDS tile and
//            12-byte (u8 RGB) coalesced stores.
// The kernels are VALU/HBM balanced (2048 unfused flops per 64 samples against 3-5 bytes/pixel).
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "kernels.hpp"

#define TILE_PITCH 72   // floats per block in LDS: 64 + 8 keeps the 8-wide row reads of neighbouring
                        // blocks on different banks

// zig-zag position -> raster index, usable in fully unrolled device code
__device__ static constexpr int kZig[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// imgproc.c:104-115 / 130-149.  in-place 2-D inverse transform, rows first then columns.
__device__ __forceinline__ void idct_block(float (&f)[64], const float (&L)[8][8])
{
	float t[64];
#pragma unroll
	for (int y = 0; y < 8; ++y) {
#pragma unroll
		for (int x = 0; x < 8; ++x) {
			float s = 0.f;
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				s += f[y * 8 + u] * L[x][u];
			}
			t[y * 8 + x] = s;
		}
	}
#pragma unroll
	for (int x = 0; x < 8; ++x) {
#pragma unroll
		for (int y = 0; y < 8; ++y) {
			float s = 0.f;
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				s += t[u * 8 + x] * L[y][u];
			}
			f[y * 8 + x] = s;
		}
	}
}

// imgproc.c:117-128 / 151-170.
__device__ __forceinline__ void fdct_block(float (&f)[64], const float (&L)[8][8])
{
	float t[64];
#pragma unroll
	for (int y = 0; y < 8; ++y) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			float s = 0.f;
#pragma unroll
			for (int x = 0; x < 8; ++x) {
				s += f[y * 8 + x] * L[x][u];
			}
			t[y * 8 + u] = s;
		}
	}
#pragma unroll
	for (int x = 0; x < 8; ++x) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			float s = 0.f;
#pragma unroll
			for (int y = 0; y < 8; ++y) {
				s += t[y * 8 + x] * L[y][u];
			}
			f[u * 8 + x] = s;
		}
	}
}

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

// ================================================================================== decode

struct DecPixArgs {
	const int16_t *coefs;      // [frame][nblk][64] zig-zag
	const JhQuant *quant;      // [frame], zig-zag order
	uint8_t *out;              // PPM/PGM bodies
	size_t frame_stride;
	int G;                     // MCUs per workgroup
	JhGeom g;
};

template <int NF>
__global__ __launch_bounds__(256) void k_decode_pixels(const DecPixArgs a)
{
	extern __shared__ __align__(16) float tile[];
	__shared__ uint16_t sq[4][64];

	const JhGeom &g = a.g;
	const int tid = threadIdx.x;
	const int mx0 = blockIdx.x * a.G, my = blockIdx.y, f = blockIdx.z;
	const int B = g.B;
	const int nmcu_here = min(a.G, g.m_x - mx0);

	for (int i = tid; i < 4 * 64; i += 256) {
		sq[i >> 6][i & 63] = a.quant[f].q[i >> 6][i & 63];
	}
	__syncthreads();

	constexpr LutHolder LH = make_lut();
	const float shiftf = (float)(1 << (g.P - 1));

	// ---- phase 1: one thread per block ------------------------------------------------
	for (int b = tid; b < nmcu_here * B; b += 256) {
		const int mcu = b / B, k = b - mcu * B;
		const int comp = g.blk_comp[k];
		const int16_t *src = a.coefs + ((size_t)f * g.nblk + ((size_t)(my * g.m_x + mx0 + mcu) * B + k)) * 64;
		const uint16_t *q = sq[g.Tq[comp] & 3];
		float v[64];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int4 w = reinterpret_cast<const int4 *>(src)[j];
			const int ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int z0 = j * 8 + e * 2;
				const int c0 = (int)(short)(ws[e] & 0xffff), c1 = ws[e] >> 16;
				// dequantize_block: int32 product, then int -> float (imgproc.c:17)
				v[kZig[z0]] = (float)(c0 * (int)q[z0]);
				v[kZig[z0 + 1]] = (float)(c1 * (int)q[z0 + 1]);
			}
		}
		idct_block(v, LH.v);
		float4 *dst = reinterpret_cast<float4 *>(tile + b * TILE_PITCH);
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			// level shift: c += shift (imgproc.c:192-194)
			dst[j] = make_float4(v[4 * j] + shiftf, v[4 * j + 1] + shiftf, v[4 * j + 2] + shiftf, v[4 * j + 3] + shiftf);
		}
	}
	__syncthreads();

	// ---- phase 2: one thread per 4 pixels ---------------------------------------------
	const int TW = a.G * 8 * g.max_H, TH = 8 * g.max_V;
	const int qpr = TW >> 2;                  // quads per tile row
	const int X = g.X, Y = g.Y;
	const int oc = NF == 1 ? 1 : 3;
	const int maxval = (1 << g.P) - 1;
	const int shift = 1 << (g.P - 1);
	const int denom = 1 << g.P;
	(void)denom;
	uint8_t *outf = a.out + (size_t)f * a.frame_stride;

	for (int qd = tid; qd < TH * qpr; qd += 256) {
		const int row = qd / qpr, px0 = (qd - row * qpr) * 4;
		const int gy = my * TH + row, gx0 = mx0 * 8 * g.max_H + px0;
		if (gy >= Y || gx0 >= X) {
			continue;
		}
		int res[4][3];
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int px = px0 + i;
			float s[NF];
#pragma unroll
			for (int c = 0; c < NF; ++c) {
				// transform_components_to_frame: frame(x, y) <- component(x / step_x, y / step_y)
				const int cx = px / g.step_x[c], cy = row / g.step_y[c];
				const int wH = 8 * g.H[c];
				const int mcu = cx / wH, cxi = cx - mcu * wH;
				const int blk = mcu * B + g.comp_first[c] + (cy >> 3) * g.H[c] + (cxi >> 3);
				s[c] = tile[blk * TILE_PITCH + (cy & 7) * 8 + (cxi & 7)];
			}
			float o[3];
			if constexpr (NF == 1) {
				o[0] = s[0];
				o[1] = o[2] = 0.f;
			} else {
				// frame.c:204-206 / 226-228: (C - shift) is a float subtract, the rest is double
				const float Yl = s[0], Cb = s[1], Cr = s[2];
				const float r_ = Yl + 1.402 * (Cr - shift);
				const float g_ = Yl - 0.34414 * (Cb - shift) - 0.71414 * (Cr - shift);
				const float b_ = Yl + 1.772 * (Cb - shift);
				if constexpr (NF == 4) {
					const float K = s[NF - 1];
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
			for (int c = 0; c < 3; ++c) {
				const int iv = (int)roundf(o[c]);               // frame.c:375-376
				res[i][c] = min(max(iv, 0), maxval);
			}
		}
		const int npx = min(4, X - gx0);
		if (g.sample_bytes == 1) {
			uint8_t *p = outf + ((size_t)gy * X + gx0) * oc;
			if (oc == 3) {
				if (npx == 4 && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
					uint32_t w0 = res[0][0] | (res[0][1] << 8) | (res[0][2] << 16) | (res[1][0] << 24);
					uint32_t w1 = res[1][1] | (res[1][2] << 8) | (res[2][0] << 16) | (res[2][1] << 24);
					uint32_t w2 = res[2][2] | (res[3][0] << 8) | (res[3][1] << 16) | (res[3][2] << 24);
					uint32_t *pw = reinterpret_cast<uint32_t *>(p);
					pw[0] = w0;
					pw[1] = w1;
					pw[2] = w2;
				} else {
					for (int i = 0; i < npx; ++i) {
						p[3 * i] = (uint8_t)res[i][0];
						p[3 * i + 1] = (uint8_t)res[i][1];
						p[3 * i + 2] = (uint8_t)res[i][2];
					}
				}
			} else {
				if (npx == 4 && ((reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
					*reinterpret_cast<uint32_t *>(p) = res[0][0] | (res[1][0] << 8) | (res[2][0] << 16) | (res[3][0] << 24);
				} else {
					for (int i = 0; i < npx; ++i) {
						p[i] = (uint8_t)res[i][0];
					}
				}
			}
		} else {
			// big-endian 16-bit samples (htons, frame.c:386)
			uint8_t *p = outf + ((size_t)gy * X + gx0) * oc * 2;
			for (int i = 0; i < npx; ++i) {
				for (int c = 0; c < oc; ++c) {
					p[(i * oc + c) * 2] = (uint8_t)(res[i][c] >> 8);
					p[(i * oc + c) * 2 + 1] = (uint8_t)res[i][c];
				}
			}
		}
	}
}

int jh_launch_decode_pixels(const JhGeom &g, int nframes, const int16_t *coefs, const JhQuant *quant,
                            void *out, size_t frame_stride, hipStream_t stream)
{
	DecPixArgs a;
	a.coefs = coefs;
	a.quant = quant;
	a.out = (uint8_t *)out;
	a.frame_stride = frame_stride;
	a.G = 32 / g.max_H > 0 ? 32 / g.max_H : 1;
	a.g = g;
	dim3 grid((g.m_x + a.G - 1) / a.G, g.m_y, nframes);
	size_t lds = (size_t)a.G * g.B * TILE_PITCH * sizeof(float);
	switch (g.Nf) {
	case 1:
		hipLaunchKernelGGL(k_decode_pixels<1>, grid, dim3(256), lds, stream, a);
		break;
	case 3:
		hipLaunchKernelGGL(k_decode_pixels<3>, grid, dim3(256), lds, stream, a);
		break;
	case 4:
		hipLaunchKernelGGL(k_decode_pixels<4>, grid, dim3(256), lds, stream, a);
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
	int G;
	JhGeom g;
};

template <int NF>
__global__ __launch_bounds__(256) void k_encode_pixels(const EncPixArgs a)
{
	extern __shared__ __align__(16) float plane[];   // NF planes of TH x TWP floats (full resolution)
	__shared__ uint16_t sq[4][64];

	const JhGeom &g = a.g;
	const int tid = threadIdx.x;
	const int mx0 = blockIdx.x * a.G, my = blockIdx.y, f = blockIdx.z;
	const int B = g.B;
	const int nmcu_here = min(a.G, g.m_x - mx0);
	const int TW = a.G * 8 * g.max_H, TH = 8 * g.max_V;
	const int TWP = TW + 1;                          // odd pitch: column walks of 8x8 gathers spread over banks
	const int X = g.X, Y = g.Y;
	const int shift = 1 << (g.P - 1);
	const uint8_t *inf = a.in + (size_t)f * a.frame_stride;

	for (int i = tid; i < 4 * 64; i += 256) {
		sq[i >> 6][i & 63] = a.quant[f].q[i >> 6][i & 63];
	}

	// ---- phase 1: samples -> float, replicate edges, RGB -> YCbCr -----------------------
	const int wpx = nmcu_here * 8 * g.max_H;
	for (int i = tid; i < TH * wpx; i += 256) {
		const int row = i / wpx, px = i - row * wpx;
		const int gx = mx0 * 8 * g.max_H + px, gy = my * TH + row;
		// read_frame_body: padding copies the last column / last row (frame.c:309-345)
		const int sx = min(gx, X - 1), sy = min(gy, Y - 1);
		float s[NF];
		if (g.sample_bytes == 1) {
			const uint8_t *p = inf + ((size_t)sy * X + sx) * NF;
#pragma unroll
			for (int c = 0; c < NF; ++c) {
				s[c] = (float)p[c];
			}
		} else {
			const uint8_t *p = inf + ((size_t)sy * X + sx) * NF * 2;
#pragma unroll
			for (int c = 0; c < NF; ++c) {
				s[c] = (float)((p[2 * c] << 8) | p[2 * c + 1]);      // ntohs (frame.c:321)
			}
		}
		if constexpr (NF == 3) if (gx < X && gy < Y) {
			// frame_to_ycc (frame.c:168-170): double arithmetic, left to right, rounded to float.
			// Padding pixels keep their raw RGB values (loops stop at X, Y: quirk Q7).
			const float R = s[0], G_ = s[1], B_ = s[2];
			const float Yl = 0.299 * R + 0.587 * G_ + 0.114 * B_;
			const float Cb = - 0.1687 * R - 0.3313 * G_ + 0.5 * B_ + shift;
			const float Cr = 0.5 * R - 0.4187 * G_ - 0.0813 * B_ + shift;
			s[0] = Yl;
			s[1] = Cb;
			s[2] = Cr;
		}
#pragma unroll
		for (int c = 0; c < NF; ++c) {
			plane[(c * TH + row) * TWP + px] = s[c];
		}
	}
	__syncthreads();

	constexpr LutHolder LH = make_lut();
	const float shiftf = (float)shift;

	// ---- phase 2: one thread per block: box mean, level shift, FDCT, quantise ----------------
	for (int b = tid; b < nmcu_here * B; b += 256) {
		const int mcu = b / B, k = b - mcu * B;
		const int comp = g.blk_comp[k];
		const int stx = g.step_x[comp], sty = g.step_y[comp];
		// top-left of the block in full-resolution tile coordinates
		const int x0 = (mcu * 8 * g.H[comp] + g.blk_h[k] * 8) * stx;
		const int y0 = (g.blk_v[k] * 8) * sty;
		const float *pl = plane + comp * TH * TWP;
		const float area = (float)(sty * stx);
		float v[64];
#pragma unroll
		for (int yy = 0; yy < 8; ++yy) {
#pragma unroll
			for (int xx = 0; xx < 8; ++xx) {
				// transform_frame_to_components (frame.c:114-123): px = 0.f; += rows then columns; /= area
				float px = 0.f;
				for (int sy = 0; sy < sty; ++sy) {
					for (int sx = 0; sx < stx; ++sx) {
						px += pl[(y0 + yy * sty + sy) * TWP + x0 + xx * stx + sx];
					}
				}
				px /= area;
				v[yy * 8 + xx] = px - shiftf;                   // forward_dct level shift (imgproc.c:220-222)
			}
		}
		fdct_block(v, LH.v);
		const uint16_t *q = sq[g.Tq[comp] & 3];
		int16_t *dst = a.coefs + ((size_t)f * g.nblk + ((size_t)(my * g.m_x + mx0 + mcu) * B + k)) * 64;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			int w[4];
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int z0 = j * 8 + e * 2;
				// quantize_block (imgproc.c:28): IEEE divide, round half away from zero
				const int c0 = (int)roundf(v[kZig[z0]] / (float)q[kZig[z0]]);
				const int c1 = (int)roundf(v[kZig[z0 + 1]] / (float)q[kZig[z0 + 1]]);
				w[e] = (c0 & 0xffff) | (c1 << 16);
			}
			reinterpret_cast<int4 *>(dst)[j] = make_int4(w[0], w[1], w[2], w[3]);
		}
	}
}

int jh_launch_encode_pixels(const JhGeom &g, int nframes, const void *in, size_t frame_stride,
                            const JhQuant *quant_raster, int16_t *coefs, hipStream_t stream)
{
	EncPixArgs a;
	a.in = (const uint8_t *)in;
	a.frame_stride = frame_stride;
	a.quant = quant_raster;
	a.coefs = coefs;
	a.G = 16 / g.max_H > 0 ? 16 / g.max_H : 1;
	a.g = g;
	dim3 grid((g.m_x + a.G - 1) / a.G, g.m_y, nframes);
	const int TW = a.G * 8 * g.max_H, TH = 8 * g.max_V;
	size_t lds = (size_t)g.Nf * TH * (TW + 1) * sizeof(float);
	switch (g.Nf) {
	case 1:
		hipLaunchKernelGGL(k_encode_pixels<1>, grid, dim3(256), lds, stream, a);
		break;
	case 3:
		hipLaunchKernelGGL(k_encode_pixels<3>, grid, dim3(256), lds, stream, a);
		break;
	default:
		return -1;
	}
	return hipGetLastError() == hipSuccess ? 0 : -1;
}
