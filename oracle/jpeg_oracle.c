/*
 * oracle/jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Scalar C99 restatement of the xbarin02/jpeg algorithms on the hot path
 * (SURVEY.md section 8a).  Works on memory buffers instead of FILE*, keeps one
 * compact component list instead of a 256-slot array, and decodes Huffman codes with
 * the canonical MINCODE/MAXCODE search instead of the reference's linear table scan;
 * the results (coefficients, bitstreams, samples) are meant to be bit-identical and
 * tests/test_oracle_vs_ref.py checks exactly that against the compiled reference.
 *
 * Floating point: build with -std=c99 -ffp-contract=off (oracle/Makefile) so that no
 * FMA is formed; every float expression below keeps the reference's evaluation order.
 *
 * Each function cites the reference file:line it follows.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "jpeg_oracle.h"

/* ======================================================================== tables */

/* zig-zag index -> raster index (common.h:34-43) */
static const uint8_t zz[64] = {
	 0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

/*
 * T1: the reference's DCT table lut[x][u] = 0.5f*C(u)*cosf((2x+1)*u*M_PI/16)
 * (imgproc.c:84-102) as evaluated by gcc 11.4/glibc 2.35, kept as IEEE-754 binary32
 * bit patterns.  It is NOT symmetric in the last ulp, so it
 * is never regenerated; test_oracle_vs_ref pins it against the compiled reference.
 */
static const uint32_t lut_bits[8][8] = {
	{0x3eb504f3, 0x3efb14be, 0x3eec835e, 0x3ed4db31, 0x3eb504f3, 0x3e8e39da, 0x3e43ef15, 0x3dc7c5c4},
	{0x3eb504f3, 0x3ed4db31, 0x3e43ef15, 0xbdc7c5c2, 0xbeb504f3, 0xbefb14bf, 0xbeec835e, 0xbe8e39dd},
	{0x3eb504f3, 0x3e8e39da, 0xbe43ef14, 0xbefb14bf, 0xbeb504f4, 0x3dc7c5c8, 0x3eec835f, 0x3ed4db32},
	{0x3eb504f3, 0x3dc7c5c4, 0xbeec835e, 0xbe8e39dd, 0x3eb504f1, 0x3ed4db32, 0xbe43ef07, 0xbefb14c0},
	{0x3eb504f3, 0xbdc7c5c2, 0xbeec835e, 0x3e8e39d7, 0x3eb504f3, 0xbed4db34, 0xbe43ef23, 0x3efb14be},
	{0x3eb504f3, 0xbe8e39d9, 0xbe43ef1a, 0x3efb14bf, 0xbeb504f0, 0xbdc7c5bb, 0x3eec8361, 0xbed4db31},
	{0x3eb504f3, 0xbed4db32, 0x3e43ef1b, 0x3dc7c5d0, 0xbeb504ef, 0x3efb14bf, 0xbeec835c, 0x3e8e39ce},
	{0x3eb504f3, 0xbefb14bf, 0x3eec835f, 0xbed4db34, 0x3eb504f4, 0xbe8e39d7, 0x3e43ef25, 0xbdc7c596}
};

static float lut[8][8];
static int lut_ready;

static void lut_init(void)
{
	if (!lut_ready) {
		memcpy(lut, lut_bits, sizeof lut);
		lut_ready = 1;
	}
}

/* Annex K.1 quantisation tables, raster order (encoder.c:14-34) */
static const uint8_t k1_luma[64] = {
	16, 11, 10, 16,  24,  40,  51,  61,   12, 12, 14, 19,  26,  58,  60,  55,
	14, 13, 16, 24,  40,  57,  69,  56,   14, 17, 22, 29,  51,  87,  80,  62,
	18, 22, 37, 56,  68, 109, 103,  77,   24, 35, 55, 64,  81, 104, 113,  92,
	49, 64, 78, 87, 103, 121, 120, 101,   72, 92, 95, 98, 112, 100, 103,  99
};

static const uint8_t k1_chroma[64] = {
	17, 18, 24, 47, 99, 99, 99, 99,   18, 21, 26, 66, 99, 99, 99, 99,
	24, 26, 56, 99, 99, 99, 99, 99,   47, 66, 99, 99, 99, 99, 99, 99,
	99, 99, 99, 99, 99, 99, 99, 99,   99, 99, 99, 99, 99, 99, 99, 99,
	99, 99, 99, 99, 99, 99, 99, 99,   99, 99, 99, 99, 99, 99, 99, 99
};

/* Annex K.3 typical Huffman tables = the reference's implicit MJPEG tables (mjpeg.h:5-82) */
static const uint8_t k3_dc_luma_bits[16]   = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k3_dc_chroma_bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k3_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k3_ac_luma_bits[16]   = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t k3_ac_chroma_bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t k3_ac_luma_vals[162] = {
	0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
	0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
	0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
	0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
	0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
	0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
	0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
	0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
	0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
	0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
	0xf9, 0xfa
};
static const uint8_t k3_ac_chroma_vals[162] = {
	0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
	0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
	0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
	0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
	0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
	0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
	0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
	0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
	0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
	0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
	0xf9, 0xfa
};

void orc_default_table(int Tc, int Th, uint8_t L[16], uint8_t V[256])
{
	memset(V, 0, 256);
	if (Tc == 0) {
		memcpy(L, Th == 0 ? k3_dc_luma_bits : k3_dc_chroma_bits, 16);
		memcpy(V, k3_dc_vals, 12);
	} else {
		memcpy(L, Th == 0 ? k3_ac_luma_bits : k3_ac_chroma_bits, 16);
		memcpy(V, Th == 0 ? k3_ac_luma_vals : k3_ac_chroma_vals, 162);
	}
}

/* ======================================================================== small helpers */

static size_t ceil_div_sz(size_t n, size_t d) { return (n + d - 1) / d; }          /* common.c:111 */
static int clampi(int lo, int v, int hi) { return v < lo ? lo : (v > hi ? hi : v); } /* common.c:197 */

void orc_free(void *p) { free(p); }

void orc_image_free(orc_image *img)
{
	if (img == NULL) {
		return;
	}
	for (int k = 0; k < ORC_MAXC; ++k) {
		free(img->comp[k].coef);
		free(img->plane[k]);
	}
	free(img->frame);
	free(img);
}

static orc_image *image_new(void)
{
	orc_image *img = calloc(1, sizeof *img);
	if (img == NULL) {
		return NULL;
	}
	/* init_qtable: all ones (common.c:10-21) */
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			img->Q[t][i] = 1;
		}
	}
	/* implicit MJPEG tables (common.c:90-99) */
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 2; ++th) {
			orc_default_table(tc, th, img->htL[tc][th], img->htV[tc][th]);
		}
	}
	return img;
}

/* geometry + coefficient buffers (common.c:156-195, alloc zero-filled common.c:129) */
static int image_alloc(orc_image *img)
{
	img->m_x = ceil_div_sz((size_t)img->X, 8u * (size_t)img->max_H);
	img->m_y = ceil_div_sz((size_t)img->Y, 8u * (size_t)img->max_V);
	img->size_x = img->m_x * 8u * (size_t)img->max_H;   /* frame.c:21 */
	img->size_y = img->m_y * 8u * (size_t)img->max_V;   /* frame.c:22 */
	for (int k = 0; k < img->ncomp; ++k) {
		orc_comp *c = &img->comp[k];
		c->bx = img->m_x * (size_t)c->H;
		c->by = img->m_y * (size_t)c->V;
		free(c->coef);
		c->coef = calloc(c->bx * c->by * 64, sizeof(int32_t));
		if (c->coef == NULL) {
			return ORC_ERR_MEMORY;
		}
	}
	return ORC_OK;
}

/* ======================================================================== Huffman codes */

/* Annex C, Figures C.1-C.3 (huffman.c:45-178) in one pass. */
typedef struct {
	int nsym;                 /* LASTK */
	uint8_t huffval[256];
	uint8_t huffsize[257];
	uint16_t huffcode[257];
	uint8_t ehufsi[256];      /* 0 = symbol absent */
	uint16_t ehufco[256];
	int32_t mincode[17], maxcode[17], valptr[17];  /* canonical decode helpers */
	int irregular;            /* more codes of some size than the size has room for (a damaged DHT): CODE, 16 bits wide in
	                           * the reference (huffman.c:91), runs past the size or wraps, the codes of a size are no
	                           * longer one range, and only query_code()'s own search says what matches */
} hufftab;

static int hufftab_build(hufftab *h, const uint8_t L[16], const uint8_t V[256])
{
	int k = 0;

	memset(h, 0, sizeof *h);
	for (int len = 1; len <= 16; ++len) {              /* C.1 */
		for (int j = 0; j < L[len - 1]; ++j) {
			if (k >= 256) {
				return ORC_ERR_LOGIC;
			}
			h->huffsize[k++] = (uint8_t)len;
		}
	}
	h->huffsize[k] = 0;
	h->nsym = k;
	memcpy(h->huffval, V, (size_t)k);

	uint32_t code = 0;                                  /* C.2 */
	int si = h->huffsize[0];
	for (int p = 0; p < k; ) {
		while (p < k && h->huffsize[p] == si) {
			if (code >> si) {
				h->irregular = 1;
			}
			h->huffcode[p++] = (uint16_t)code++;
		}
		code = (code << 1) & 0x1ffff;                   /* (bit 16 kept: "past the size" must stay visible) */
		si++;
	}

	for (int len = 1, p = 0; len <= 16; ++len) {        /* decode helpers (F.2.2.3 style) */
		if (L[len - 1]) {
			h->valptr[len] = p;
			h->mincode[len] = h->huffcode[p];
			p += L[len - 1];
			h->maxcode[len] = h->huffcode[p - 1];
		} else {
			h->maxcode[len] = -1;
		}
	}

	for (int p = 0; p < k; ++p) {                       /* C.3 (later duplicates overwrite) */
		h->ehufco[h->huffval[p]] = h->huffcode[p];
		h->ehufsi[h->huffval[p]] = h->huffsize[p];
	}
	return ORC_OK;
}

int orc_make_codes(const uint8_t L[16], const uint8_t V[256], uint8_t ehufsi[256], uint16_t ehufco[256])
{
	hufftab h;
	int err = hufftab_build(&h, L, V);
	if (err) {
		return err;
	}
	/*
	 * The reference encodes with value_to_vlc() (huffman.c:228): FIRST table entry whose
	 * HUFFVAL equals the symbol.  For tables without duplicate symbols this is EHUFCO/EHUFSI.
	 */
	memset(ehufsi, 0, 256);
	memset(ehufco, 0, 512);
	for (int p = h.nsym - 1; p >= 0; --p) {
		ehufco[h.huffval[p]] = h.huffcode[p];
		ehufsi[h.huffval[p]] = h.huffsize[p];
	}
	return ORC_OK;
}

/* ======================================================================== K.2 optimiser */

/* huffman.c:330-364: least FREQ > 0, ties -> LARGEST index ("<=" scan upwards) */
static int k2_least(const size_t freq[257], int exclude)
{
	size_t best = (size_t)-1;
	int v = -1;
	for (int i = 0; i < 257; ++i) {
		if (freq[i] > 0 && freq[i] <= best && i != exclude) {
			best = freq[i];
			v = i;
		}
	}
	return v;
}

void orc_adapt_table(const size_t freq_in[257], uint8_t L[16], uint8_t V[256])
{
	size_t freq[257];
	int codesize[257];
	int others[257];
	int bits[33];

	memcpy(freq, freq_in, sizeof freq);
	for (int i = 0; i < 257; ++i) {
		codesize[i] = 0;
		others[i] = -1;
	}

	/* Figure K.1 (huffman.c:366-411) */
	for (;;) {
		int v1 = k2_least(freq, -1);
		int v2 = k2_least(freq, v1);
		if (v2 < 0) {
			break;
		}
		freq[v1] += freq[v2];
		freq[v2] = 0;
		for (;;) {
			codesize[v1]++;
			if (others[v1] < 0) {
				break;
			}
			v1 = others[v1];
		}
		others[v1] = v2;
		for (;;) {
			codesize[v2]++;
			if (others[v2] < 0) {
				break;
			}
			v2 = others[v2];
		}
	}

	/* Figure K.2 (huffman.c:446-474) */
	memset(bits, 0, sizeof bits);
	for (int i = 0; i < 257; ++i) {
		if (codesize[i] != 0 && codesize[i] < 33) {
			bits[codesize[i]]++;
		}
	}

	/* Figure K.3 (huffman.c:413-444) */
	for (int i = 32; i > 16; ) {
		if (bits[i] > 0) {
			int j = i - 1;
			do {
				j--;
			} while (bits[j] <= 0);
			bits[i] -= 2;
			bits[i - 1] += 1;
			bits[j + 1] += 2;
			bits[j] -= 1;
		} else {
			i--;
		}
	}
	{
		int i = 16;
		while (bits[i] == 0) {
			i--;
		}
		bits[i] -= 1;   /* drop the reserved all-ones code point */
	}

	/* Figure K.4 (huffman.c:476-506): symbols 0..255 ascending per code size */
	memset(V, 0, 256);
	int k = 0;
	for (int i = 1; i <= 32; ++i) {
		for (int j = 0; j <= 255; ++j) {
			if (codesize[j] == i && k < 256) {
				V[k++] = (uint8_t)j;
			}
		}
	}
	for (int i = 0; i < 16; ++i) {
		L[i] = (uint8_t)bits[i + 1];
	}
}

/* ======================================================================== coefficients */

int orc_encode_cat(int32_t c)          /* coeffs.c:56-74 */
{
	int r = 0;
	if (c < 0) {
		c = -c;
	}
	while (c != 0) {
		c >>= 1;
		r++;
	}
	return r;
}

int orc_encode_extra(int32_t c, int cat)   /* coeffs.c:76-83 */
{
	if (c < 0) {
		c--;
	}
	return (int)((uint32_t)c & ((UINT32_C(1) << cat) - 1)) & 0xffff;
}

int32_t orc_decode_coeff(int cat, int extra)   /* coeffs.c:33-48 (Figure F.12) */
{
	if (cat == 0) {
		return 0;
	}
	if ((extra >> (cat - 1)) == 0) {
		return -(INT32_C(1) << cat) + extra + 1;
	}
	return extra;
}

/* ======================================================================== DCT / quant */

void orc_lut(float out[64])
{
	lut_init();
	memcpy(out, lut, sizeof lut);
}

/* imgproc.c:104-115: out[x] = sum_u in[u]*lut[x][u], s starts at 0.f, u ascending */
static void idct_1d(const float *in, float *out, size_t stride)
{
	for (int x = 0; x < 8; ++x) {
		float s = 0.f;
		for (int u = 0; u < 8; ++u) {
			s += in[(size_t)u * stride] * lut[x][u];
		}
		out[(size_t)x * stride] = s;
	}
}

/* imgproc.c:117-128: out[u] = sum_x in[x]*lut[x][u] */
static void fdct_1d(const float *in, float *out, size_t stride)
{
	for (int u = 0; u < 8; ++u) {
		float s = 0.f;
		for (int x = 0; x < 8; ++x) {
			s += in[(size_t)x * stride] * lut[x][u];
		}
		out[(size_t)u * stride] = s;
	}
}

void orc_idct_block(float c[64])   /* imgproc.c:130-149: rows, then columns */
{
	float t[64];
	lut_init();
	for (int y = 0; y < 8; ++y) {
		idct_1d(&c[y * 8], &t[y * 8], 1);
	}
	for (int x = 0; x < 8; ++x) {
		idct_1d(&t[x], &c[x], 8);
	}
}

void orc_fdct_block(float c[64])   /* imgproc.c:151-170 */
{
	float t[64];
	lut_init();
	for (int y = 0; y < 8; ++y) {
		fdct_1d(&c[y * 8], &t[y * 8], 1);
	}
	for (int x = 0; x < 8; ++x) {
		fdct_1d(&t[x], &c[x], 8);
	}
}

void orc_quantize_block(int32_t out[64], const float in[64], const uint16_t Q[64])   /* imgproc.c:21-30 */
{
	for (int j = 0; j < 64; ++j) {
		out[j] = (int32_t)roundf(in[j] / (float)Q[j]);
	}
}

void orc_dequantize_block(float out[64], const int32_t in[64], const uint16_t Q[64]) /* imgproc.c:10-19 */
{
	for (int j = 0; j < 64; ++j) {
		out[j] = (float)(in[j] * (int32_t)Q[j]);
	}
}

void orc_set_qtable(uint16_t Q[64], int chroma, int q)   /* encoder.c:38-65 */
{
	const uint8_t *ref = chroma ? k1_chroma : k1_luma;
	int sf;
	if (q < 1) {
		q = 1;
	}
	if (q > 100) {
		q = 100;
	}
	sf = q < 50 ? 5000 / q : 200 - q * 2;
	for (int i = 0; i < 64; ++i) {
		Q[i] = (uint16_t)clampi(1, (int)(((unsigned)ref[i] * (unsigned)sf + 50u) / 100u), 255);
	}
}

/* ======================================================================== byte sink */

typedef struct {
	uint8_t *p;
	size_t n, cap;
	int oom;
} sink;

static void sink_put(sink *s, uint8_t b)
{
	if (s->n == s->cap) {
		size_t nc = s->cap ? s->cap * 2 : 4096;
		uint8_t *np = realloc(s->p, nc);
		if (np == NULL) {
			s->oom = 1;
			return;
		}
		s->p = np;
		s->cap = nc;
	}
	s->p[s->n++] = b;
}

static void sink_word(sink *s, unsigned w) { sink_put(s, (uint8_t)(w >> 8)); sink_put(s, (uint8_t)w); }
static void sink_marker(sink *s, unsigned m) { sink_put(s, 0xff); sink_put(s, (uint8_t)m); }

/* bit writer: MSB first, FF -> FF 00, final pad with 1s (io.c:43-87, 277-290) */
typedef struct {
	sink *s;
	unsigned acc;
	int cnt;
} bitw;

static void bitw_byte(bitw *w, unsigned b)
{
	sink_put(w->s, (uint8_t)b);
	if ((b & 0xff) == 0xff) {
		sink_put(w->s, 0x00);
	}
}

static void bitw_put(bitw *w, unsigned value, int nbits)
{
	for (int s = nbits - 1; s >= 0; --s) {
		w->acc = (w->acc << 1) | ((value >> s) & 1u);
		if (++w->cnt == 8) {
			bitw_byte(w, w->acc & 0xff);
			w->acc = 0;
			w->cnt = 0;
		}
	}
}

static void bitw_flush(bitw *w)
{
	if (w->cnt == 0) {
		return;
	}
	while (w->cnt < 8) {
		w->acc = (w->acc << 1) | 1u;
		w->cnt++;
	}
	bitw_byte(w, w->acc & 0xff);
	w->acc = 0;
	w->cnt = 0;
}

/* ======================================================================== bit source */

/* entropy-coded segment reader on memory (io.c:18-41, 247-274) */
typedef struct {
	const uint8_t *p;
	size_t n, pos;
	unsigned byte;
	int cnt;
} bitr;

/* returns 0/1, or -1 when the segment ends (FF followed by non-00) or data runs out */
static int bitr_bit(bitr *r)
{
	if (r->cnt == 0) {
		if (r->pos >= r->n) {
			return -1;   /* RET_FAILURE_FILE_IO in the reference; both end the scan loop... */
		}
		unsigned b = r->p[r->pos];
		if (b == 0xff) {
			if (r->pos + 1 >= r->n) {
				return -1;
			}
			if (r->p[r->pos + 1] != 0x00) {
				return -1;   /* marker: stream stays positioned ON it (io.c:265) */
			}
			r->pos += 2;
		} else {
			r->pos += 1;
		}
		r->byte = b;
		r->cnt = 8;
	}
	int bit = (r->byte >> 7) & 1;
	r->byte = (r->byte << 1) & 0xff;
	r->cnt--;
	return bit;
}

/* huffman.c:255-271 read_code, with the linear search replaced by the canonical one */
static int read_symbol(bitr *r, const hufftab *h)
{
	int32_t code = 0;
	for (int len = 1; len <= 16; ++len) {
		int b = bitr_bit(r);
		if (b < 0) {
			return -1;
		}
		code = (code << 1) | b;
		if (h->irregular) {
			/* query_code() as it is (huffman.c:193-225): the first entry of this size with this code */
			for (int p = 0; p < h->nsym; ++p) {
				if (h->huffsize[p] == len && h->huffcode[p] == (uint16_t)code) {
					return h->huffval[p];
				}
			}
			continue;
		}
		if (h->maxcode[len] >= 0 && code >= h->mincode[len] && code <= h->maxcode[len]) {
			return h->huffval[h->valptr[len] + (code - h->mincode[len])];
		}
	}
	/* the reference keeps consuming bits until the data ends (vlc.size never matches again) */
	while (bitr_bit(r) >= 0) {
	}
	return -1;
}

static int read_extra(bitr *r, int count, int *value)   /* huffman.c:294-312 */
{
	unsigned v = 0;
	for (int i = 0; i < count; ++i) {
		int b = bitr_bit(r);
		if (b < 0) {
			return -1;
		}
		v = ((v << 1) | (unsigned)b) & 0xffff;
	}
	*value = (int)v;
	return 0;
}

/* coeffs.c:196-254 read_block; blk == NULL marks "past the end" (decoder.c:342-344) */
static int read_block(bitr *r, const hufftab *dc, const hufftab *ac, int32_t *blk)
{
	int sym, extra;

	sym = read_symbol(r, dc);
	if (sym < 0 || read_extra(r, sym, &extra) < 0) {
		return ORC_ERR_NO_MORE_DATA;
	}
	if (blk == NULL) {
		return ORC_ERR_NO_MORE_DATA;
	}
	memset(blk, 0, 64 * sizeof(int32_t));
	blk[0] = orc_decode_coeff(sym & 0xff, extra);

	int i = 1, rem = 63;
	do {
		sym = read_symbol(r, ac);
		if (sym < 0) {
			return ORC_ERR_NO_MORE_DATA;
		}
		int cat = sym & 15, run = sym >> 4;
		if (read_extra(r, cat, &extra) < 0) {
			return ORC_ERR_NO_MORE_DATA;
		}
		if (sym == 0) {
			break;   /* EOB */
		}
		i += run;
		if (i <= 63) {
			blk[zz[i]] = orc_decode_coeff(cat, extra);
		}
		/* i > 63 (a zero run past the end of the block, damaged data): the reference reads zigzag[] out of
		 * range and stores the value wherever that byte points (undefined behaviour, coeffs.c:246-247); its
		 * CONTROL FLOW is defined all the same: rem drops to 0 or below, the block is complete and decoding
		 * goes on with the next one.  The store is left out here. */
		i++;
		rem -= run + 1;
	} while (rem > 0);

	return ORC_OK;
}

/* ======================================================================== decoder */

typedef struct {
	int Ns;
	int comp[ORC_MAXC];        /* index into img->comp, scan order */
	int32_t *last[ORC_MAXC];   /* previous block of that component (DC predictor) */
} scan_t;

static int find_comp(const orc_image *img, int id)
{
	for (int k = 0; k < img->ncomp; ++k) {
		if (img->comp[k].id == id) {
			return k;
		}
	}
	return -1;
}

/* decoder.c:262-362 read_macroblock */
static int read_mcu(bitr *r, orc_image *img, scan_t *sc, const hufftab ht[2][4])
{
	size_t seq = img->mblocks;

	if (sc->Ns == 0) {
		return ORC_ERR_NO_MORE_DATA;
	}
	if (sc->Ns == 1) {
		orc_comp *c = &img->comp[sc->comp[0]];
		size_t nb = (size_t)c->H * (size_t)c->V;
		for (size_t w = 0; w < nb; ++w) {
			size_t b = nb * seq + w;   /* plain raster over the padded b_x (decoder.c:285-289) */
			if (b >= c->bx * c->by) {
				return ORC_ERR_NO_MORE_DATA;   /* reference: out-of-bounds write */
			}
			int32_t *blk = &c->coef[b * 64];
			int err = read_block(r, &ht[0][c->Td], &ht[1][c->Ta], blk);
			if (err) {
				return err;
			}
			if (sc->last[0] != NULL) {
				blk[0] += sc->last[0][0];
			}
			sc->last[0] = blk;
		}
		return ORC_OK;
	}
	if (img->m_x == 0) {
		return ORC_ERR_UNSUPPORTED;
	}
	size_t x = seq % img->m_x, y = seq / img->m_x;
	for (int j = 0; j < sc->Ns; ++j) {
		orc_comp *c = &img->comp[sc->comp[j]];
		for (int v = 0; v < c->V; ++v) {
			for (int h = 0; h < c->H; ++h) {
				size_t b = (y * (size_t)c->V + (size_t)v) * c->bx + x * (size_t)c->H + (size_t)h;
				int32_t *blk = b < c->bx * c->by ? &c->coef[b * 64] : NULL;
				int err = read_block(r, &ht[0][c->Td], &ht[1][c->Ta], blk);
				if (err) {
					return err;
				}
				if (sc->last[j] != NULL) {
					blk[0] += sc->last[j][0];
				}
				sc->last[j] = blk;
			}
		}
	}
	return ORC_OK;
}

/* decoder.c:364-388 read_ecs: one entropy-coded segment, returns the new position */
static int read_ecs(const uint8_t *p, size_t n, size_t *pos, orc_image *img, scan_t *sc,
                    const hufftab ht[2][4])
{
	bitr r = {p, n, *pos, 0, 0};
	for (int j = 0; j < ORC_MAXC; ++j) {
		sc->last[j] = NULL;
	}
	for (;;) {
		int err = read_mcu(&r, img, sc, ht);
		if (err == ORC_ERR_NO_MORE_DATA) {
			break;
		}
		if (err) {
			return err;
		}
		img->mblocks++;
	}
	*pos = r.pos;
	return ORC_OK;
}

/* frame.c:38-82 + 134-152: replicate-upsample the component planes into one interleaved frame */
static void planes_to_frame(const orc_image *img, float *const plane[ORC_MAXC], float *frame)
{
	size_t size_x = img->size_x, size_y = img->size_y;
	size_t nf = (size_t)img->Nf;
	for (int k = 0; k < img->ncomp; ++k) {
		const orc_comp *c = &img->comp[k];
		size_t cx = c->bx * 8, cy = c->by * 8;
		size_t sx = size_x / cx, sy = size_y / cy;
		for (size_t y = 0; y < cy; ++y) {
			for (size_t x = 0; x < cx; ++x) {
				float px = plane[k][y * cx + x];
				for (size_t yy = 0; yy < sy; ++yy) {
					for (size_t xx = 0; xx < sx; ++xx) {
						frame[(sy * y + yy) * size_x * nf + nf * (sx * x + xx) + (size_t)k] = px;
					}
				}
			}
		}
	}
}

/* frame.c:188-244 frame_to_rgb */
static int frame_to_rgb(const orc_image *img, float *d)
{
	int shift = 1 << (img->P - 1);
	int denom = 1 << img->P;
	size_t sx = img->size_x;

	switch (img->Nf) {
	case 4:
		for (size_t y = 0; y < (size_t)img->Y; ++y) {
			for (size_t x = 0; x < (size_t)img->X; ++x) {
				float *px = &d[y * sx * 4 + x * 4];
				float Y_ = px[0], Cb = px[1], Cr = px[2], K = px[3];
				float C = Y_ + 1.402 * (Cr - shift);
				float M = Y_ - 0.34414 * (Cb - shift) - 0.71414 * (Cr - shift);
				float Yl = Y_ + 1.772 * (Cb - shift);
				px[0] = K - (C * K) / denom;
				px[1] = K - (M * K) / denom;
				px[2] = K - (Yl * K) / denom;
				px[3] = 0xff;
			}
		}
		break;
	case 3:
		for (size_t y = 0; y < (size_t)img->Y; ++y) {
			for (size_t x = 0; x < (size_t)img->X; ++x) {
				float *px = &d[y * sx * 3 + x * 3];
				float Yl = px[0], Cb = px[1], Cr = px[2];
				float R = Yl + 1.402 * (Cr - shift);
				float G = Yl - 0.34414 * (Cb - shift) - 0.71414 * (Cr - shift);
				float B = Yl + 1.772 * (Cb - shift);
				px[0] = R;
				px[1] = G;
				px[2] = B;
			}
		}
		break;
	case 1:
		break;
	default:
		return ORC_ERR_UNSUPPORTED;   /* the reference abort()s */
	}
	return ORC_OK;
}

/* decoder.c:456-470 epilogue + write_image, producing the PPM/PGM body in memory */
static int epilogue(orc_image *img, uint8_t **samples, size_t *samples_len)
{
	float *plane[ORC_MAXC] = {0};
	int err = ORC_OK;
	int shift = 1 << (img->P - 1);

	if (img->ncomp == 0 || img->X == 0 || img->Y == 0) {
		return ORC_ERR_UNSUPPORTED;
	}
	lut_init();
	for (int k = 0; k < img->ncomp; ++k) {
		const orc_comp *c = &img->comp[k];
		size_t nblk = c->bx * c->by;
		const uint16_t *Q = img->Q[c->Tq & 3];
		plane[k] = malloc(nblk * 64 * sizeof(float));
		if (plane[k] == NULL) {
			err = ORC_ERR_MEMORY;
			goto out;
		}
		for (size_t by = 0; by < c->by; ++by) {
			for (size_t bx = 0; bx < c->bx; ++bx) {
				float f[64];
				orc_dequantize_block(f, &c->coef[(by * c->bx + bx) * 64], Q);   /* imgproc.c:32-56 */
				orc_idct_block(f);                                             /* imgproc.c:172-200 */
				for (int j = 0; j < 64; ++j) {
					f[j] += shift;
				}
				for (int v = 0; v < 8; ++v) {                                  /* imgproc.c:233-262 */
					for (int u = 0; u < 8; ++u) {
						plane[k][(by * 8 + (size_t)v) * (c->bx * 8) + bx * 8 + (size_t)u] = f[v * 8 + u];
					}
				}
			}
		}
	}

	free(img->frame);
	img->frame = malloc(sizeof(float) * (size_t)img->Nf * img->size_x * img->size_y);
	if (img->frame == NULL) {
		err = ORC_ERR_MEMORY;
		goto out;
	}
	/* malloc()ed and only partly written in the reference; zero here for determinism */
	memset(img->frame, 0, sizeof(float) * (size_t)img->Nf * img->size_x * img->size_y);
	planes_to_frame(img, plane, img->frame);
	err = frame_to_rgb(img, img->frame);
	if (err) {
		goto out;
	}

	/* frame.c:352-405 write_frame_body: roundf, (int), clamp, u8 / big-endian u16 */
	{
		int maxval = (1 << img->P) - 1;
		int oc = img->Nf == 1 ? 1 : 3;
		size_t ss = maxval <= 255 ? 1 : 2;
		size_t nf = (size_t)img->Nf;
		size_t len = ss * (size_t)oc * (size_t)img->X * (size_t)img->Y;
		uint8_t *o = malloc(len ? len : 1), *w;
		if (o == NULL) {
			err = ORC_ERR_MEMORY;
			goto out;
		}
		w = o;
		for (size_t y = 0; y < (size_t)img->Y; ++y) {
			for (size_t x = 0; x < (size_t)img->X; ++x) {
				for (int c = 0; c < oc; ++c) {
					float s = roundf(img->frame[y * img->size_x * nf + x * nf + (size_t)c]);
					int v = clampi(0, (int)s, maxval);
					if (ss == 1) {
						*w++ = (uint8_t)v;
					} else {
						*w++ = (uint8_t)(v >> 8);
						*w++ = (uint8_t)v;
					}
				}
			}
		}
		*samples = o;
		*samples_len = len;
	}
out:
	for (int k = 0; k < ORC_MAXC; ++k) {
		free(plane[k]);
	}
	return err;
}

/* io.c:186-220 read_marker on memory */
static int next_marker(const uint8_t *p, size_t n, size_t *pos, unsigned *marker)
{
	size_t i = *pos;
	for (;;) {
		do {
			if (i >= n) {
				return ORC_ERR_IO;
			}
		} while (p[i++] != 0xff);
		for (;;) {
			if (i >= n) {
				return ORC_ERR_IO;
			}
			unsigned b = p[i++];
			if (b == 0xff) {
				continue;
			}
			if (b == 0x00) {
				break;   /* not a marker: keep seeking */
			}
			*marker = 0xff00u | b;
			*pos = i;
			return ORC_OK;
		}
	}
}

#define NEED(k) do { if (pos + (size_t)(k) > n) { err = ORC_ERR_IO; goto fail; } } while (0)

/* decoder.c:472-659 parse_format + 661-685 process_jpeg_stream */
int orc_decode(const uint8_t *jpg, size_t n, orc_image **img_out, uint8_t **samples, size_t *samples_len)
{
	orc_image *img = image_new();
	hufftab (*ht)[4] = calloc(2, sizeof(hufftab[4]));
	scan_t sc;
	size_t pos = 0;
	int err = ORC_OK;

	*samples = NULL;
	*samples_len = 0;
	if (img_out) {
		*img_out = NULL;
	}
	if (img == NULL || ht == NULL) {
		err = ORC_ERR_MEMORY;
		goto fail;
	}
	memset(&sc, 0, sizeof sc);
	/* Table ids 2 and 3 with no codes until a DHT segment defines them.  The reference's init_context() never
	 * fills context->hcode[][2..3] (common.c:90-99 converts the four implicit tables only; the context is
	 * malloc()ed, decoder.c:665): a scan that names such a table without a DHT reads uninitialised memory there,
	 * and what it decodes changes from run to run.  On fresh (zero) pages query_code() matches nothing (last_k = 0,
	 * huffman.c:203-216), read_code() runs on to the end of the data (huffman.c:262-268) -- that reading is taken
	 * here, and by the HIP path. */
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 4; ++th) {
			hufftab_build(&ht[tc][th], img->htL[tc][th], img->htV[tc][th]);
		}
	}

	for (;;) {
		unsigned marker;
		err = next_marker(jpg, n, &pos, &marker);
		if (err) {
			goto fail;
		}
		if (marker == 0xffd8 || marker == 0xff01) {           /* SOI, TEM */
			continue;
		}
		if ((marker >= 0xffe0 && marker <= 0xffe8) || (marker >= 0xffeb && marker <= 0xffee) ||
		    marker == 0xffcc) {                                 /* APPn (not 9, 10, 15), DAC: skipped */
			NEED(2);
			size_t len = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
			pos += len;   /* read_length + skip_segment(len - 2) */
			continue;
		}
		if (marker == 0xfffe) {                                 /* COM */
			NEED(2);
			size_t len = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
			if (len < 2) {
				err = ORC_ERR_UNSUPPORTED;
				goto fail;
			}
			NEED(len);
			pos += len;
			continue;
		}
		if (marker == 0xffdb) {                                 /* DQT (decoder.c:20-68, 518-527) */
			size_t seg = pos;
			NEED(2);
			size_t len = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
			pos += 2;
			do {
				NEED(1);
				int Pq = jpg[pos] >> 4, Tq = jpg[pos] & 15;
				pos++;
				if (Tq >= 4 || Pq >= 2) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				NEED(Pq ? 128 : 64);
				for (int i = 0; i < 64; ++i) {
					if (Pq == 0) {
						img->Q[Tq][zz[i]] = jpg[pos++];
					} else {
						img->Q[Tq][zz[i]] = (uint16_t)((jpg[pos] << 8) | jpg[pos + 1]);
						pos += 2;
					}
				}
			} while (pos < seg + len);
			continue;
		}
		if (marker == 0xffc0 || marker == 0xffc1) {             /* SOF0 / SOF1 (decoder.c:70-136) */
			NEED(8);
			pos += 2;
			img->P = jpg[pos];
			img->Y = (jpg[pos + 1] << 8) | jpg[pos + 2];
			img->X = (jpg[pos + 3] << 8) | jpg[pos + 4];
			img->Nf = jpg[pos + 5];
			pos += 6;
			if (img->X == 0 || img->Nf == 0 || img->Nf > ORC_MAXC || img->P < 2 || img->P > 16) {
				err = ORC_ERR_UNSUPPORTED;
				goto fail;
			}
			NEED(3 * img->Nf);
			img->max_H = img->max_V = 0;
			for (int i = 0; i < img->Nf; ++i) {
				int id = jpg[pos], H = jpg[pos + 1] >> 4, V = jpg[pos + 1] & 15, Tq = jpg[pos + 2];
				pos += 3;
				int k = find_comp(img, id);
				if (k < 0) {
					/* keep the list in ascending id order, as the reference's 0..255 walk */
					k = img->ncomp++;
					while (k > 0 && img->comp[k - 1].id > id) {
						img->comp[k] = img->comp[k - 1];
						k--;
					}
					memset(&img->comp[k], 0, sizeof img->comp[k]);
				}
				if (H == 0 || V == 0 || Tq >= 4) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				img->comp[k].id = id;
				img->comp[k].H = H;
				img->comp[k].V = V;
				img->comp[k].Tq = Tq;
				img->max_H = H > img->max_H ? H : img->max_H;
				img->max_V = V > img->max_V ? V : img->max_V;
			}
			err = image_alloc(img);
			if (err) {
				goto fail;
			}
			continue;
		}
		if (marker == 0xffc2 || marker == 0xffc3 || marker == 0xffc9 || marker == 0xffca) {
			err = ORC_ERR_UNSUPPORTED;                          /* decoder.c:545-579 */
			goto fail;
		}
		if (marker == 0xffc4) {                                 /* DHT (decoder.c:143-184, 581-591) */
			size_t seg = pos;
			NEED(2);
			size_t len = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
			pos += 2;
			do {
				NEED(17);
				int Tc = jpg[pos] >> 4, Th = jpg[pos] & 15;
				pos++;
				if (Tc >= 2 || Th >= 4) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				size_t total = 0;
				for (int i = 0; i < 16; ++i) {
					img->htL[Tc][Th][i] = jpg[pos + (size_t)i];
					total += jpg[pos + (size_t)i];
				}
				pos += 16;
				if (total > 256) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				NEED(total);
				memset(img->htV[Tc][Th], 0, 256);
				memcpy(img->htV[Tc][Th], &jpg[pos], total);
				pos += total;
				err = hufftab_build(&ht[Tc][Th], img->htL[Tc][Th], img->htV[Tc][Th]);
				if (err) {
					goto fail;
				}
			} while (pos < seg + len);
			continue;
		}
		if (marker == 0xffdd) {                                 /* DRI (decoder.c:390-401) */
			NEED(4);
			img->Ri = (jpg[pos + 2] << 8) | jpg[pos + 3];      /* parsed, never used */
			pos += 4;
			continue;
		}
		if (marker == 0xffda) {                                 /* SOS (decoder.c:197-259) */
			NEED(3);
			pos += 2;
			sc.Ns = jpg[pos++];
			if (sc.Ns > ORC_MAXC) {
				err = ORC_ERR_UNSUPPORTED;
				goto fail;
			}
			NEED(2 * sc.Ns + 3);
			for (int j = 0; j < sc.Ns; ++j) {
				int k = find_comp(img, jpg[pos]);
				if (k < 0) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				sc.comp[j] = k;
				img->comp[k].Td = jpg[pos + 1] >> 4;
				img->comp[k].Ta = jpg[pos + 1] & 15;
				if (img->comp[k].Td >= 4 || img->comp[k].Ta >= 4) {
					err = ORC_ERR_UNSUPPORTED;
					goto fail;
				}
				pos += 2;
			}
			if (jpg[pos] != 0 || jpg[pos + 1] != 63 || jpg[pos + 2] != 0) {
				err = ORC_ERR_UNSUPPORTED;
				goto fail;
			}
			pos += 3;
			img->mblocks = 0;
			err = read_ecs(jpg, n, &pos, img, &sc, (const hufftab (*)[4])ht);
			if (err) {
				goto fail;
			}
			continue;
		}
		if (marker >= 0xffd0 && marker <= 0xffd7) {             /* RSTm (decoder.c:622-633) */
			err = read_ecs(jpg, n, &pos, img, &sc, (const hufftab (*)[4])ht);
			if (err) {
				goto fail;
			}
			continue;
		}
		if (marker == 0xffd9) {                                 /* EOI (decoder.c:603-612) */
			err = epilogue(img, samples, samples_len);
			if (err) {
				goto fail;
			}
			break;
		}
		err = ORC_ERR_UNSUPPORTED;                              /* decoder.c:654-656 */
		goto fail;
	}

	free(ht);
	if (img_out) {
		*img_out = img;
	} else {
		orc_image_free(img);
	}
	return ORC_OK;
fail:
	free(ht);
	orc_image_free(img);
	return err;
}

/* ======================================================================== encoder */

/* coeffs.c:256-309 write_block (dry == NULL) / coeffs.c:312-363 write_block_dry (freq != NULL) */
static int code_block(bitw *w, const hufftab *dc, const hufftab *ac, size_t *fdc, size_t *fac,
                      const int32_t *blk)
{
	int cat = orc_encode_cat(blk[0]);

	if (fdc) {
		fdc[cat]++;
	} else {
		if (dc->ehufsi[cat] == 0) {
			return ORC_ERR_NO_CODE;
		}
		bitw_put(w, dc->ehufco[cat], dc->ehufsi[cat]);
		bitw_put(w, (unsigned)orc_encode_extra(blk[0], cat), cat);
	}
	for (int r = 0, i = 1; i < 64; ++i) {
		int32_t c = blk[zz[i]];
		if (c == 0) {
			if (i == 63) {
				if (fac) {
					fac[0]++;
				} else {
					if (ac->ehufsi[0] == 0) {
						return ORC_ERR_NO_CODE;
					}
					bitw_put(w, ac->ehufco[0], ac->ehufsi[0]);   /* EOB */
				}
			} else {
				r++;
			}
			continue;
		}
		while (r > 15) {                                        /* ZRL */
			if (fac) {
				fac[0xf0]++;
			} else {
				if (ac->ehufsi[0xf0] == 0) {
					return ORC_ERR_NO_CODE;
				}
				bitw_put(w, ac->ehufco[0xf0], ac->ehufsi[0xf0]);
			}
			r -= 16;
		}
		cat = orc_encode_cat(c);
		int rs = ((r & 15) << 4) | (cat & 15);
		if (fac) {
			fac[rs]++;
		} else {
			if (ac->ehufsi[rs] == 0) {
				return ORC_ERR_NO_CODE;
			}
			bitw_put(w, ac->ehufco[rs], ac->ehufsi[rs]);
			bitw_put(w, (unsigned)orc_encode_extra(c, cat), cat);
		}
		r = 0;
	}
	return ORC_OK;
}

/*
 * encoder.c:412-464 + 560-587 (write) and 466-558 (dry): MCU walk with DC prediction.
 * freq != NULL selects the dry run.  ri > 0 (extension) restarts every ri MCUs.
 */
static int walk_mcus(orc_image *img, sink *s, hufftab enc[2][4], size_t (*freq)[4][257], int ri)
{
	bitw w = {s, 0, 0};
	int32_t pred[ORC_MAXC] = {0};
	size_t total = img->m_x * img->m_y;
	unsigned rst = 0;

	for (size_t seq = 0; seq < total; ++seq) {
		size_t x = seq % img->m_x, y = seq / img->m_x;
		if (ri > 0 && seq > 0 && seq % (size_t)ri == 0) {
			if (!freq) {
				bitw_flush(&w);
				sink_marker(s, 0xd0 + (rst & 7));
			}
			rst++;
			memset(pred, 0, sizeof pred);
		}
		for (int k = 0; k < img->ncomp; ++k) {
			orc_comp *c = &img->comp[k];
			for (int v = 0; v < c->V; ++v) {
				for (int h = 0; h < c->H; ++h) {
					size_t b = (y * (size_t)c->V + (size_t)v) * c->bx + x * (size_t)c->H + (size_t)h;
					int32_t blk[64];
					memcpy(blk, &c->coef[b * 64], sizeof blk);
					int32_t dcv = blk[0];
					blk[0] -= pred[k];
					pred[k] = dcv;
					if (blk[0] < -2047 || blk[0] > 2047) {
						return ORC_ERR_DC_RANGE;   /* assert in the reference (encoder.c:447) */
					}
					int err = code_block(&w, &enc[0][c->Td], &enc[1][c->Ta],
					                     freq ? freq[0][c->Td] : NULL, freq ? freq[1][c->Ta] : NULL, blk);
					if (err) {
						return err;
					}
				}
			}
		}
	}
	if (!freq) {
		bitw_flush(&w);
	}
	img->mblocks = total;
	return ORC_OK;
}

static void put_dht(sink *s, int Tc, int Th, const uint8_t L[16], const uint8_t V[256])   /* encoder.c:276-316 */
{
	unsigned mt = 0;
	for (int i = 0; i < 16; ++i) {
		mt += L[i];
	}
	sink_marker(s, 0xc4);
	sink_word(s, 2 + 17 + mt);
	sink_put(s, (uint8_t)((Tc << 4) | Th));
	for (int i = 0; i < 16; ++i) {
		sink_put(s, L[i]);
	}
	for (unsigned i = 0; i < mt; ++i) {
		sink_put(s, V[i]);
	}
}

/* frame.c:162-176, the body of frame_to_ycc's loops: n interleaved (R, G, B) float triples in place.
 * Binary64 arithmetic, left to right, rounded to binary32 on assignment. */
void orc_frame_to_ycc(float *px, size_t n, int P)
{
	int shift = 1 << (P - 1);
	for (size_t i = 0; i < n; ++i, px += 3) {
		float R = px[0], G = px[1], B = px[2];
		float Yl = 0.299 * R + 0.587 * G + 0.114 * B;
		float Cb = - 0.1687 * R - 0.3313 * G + 0.5 * B + shift;
		float Cr = 0.5 * R - 0.4187 * G - 0.0813 * B + shift;
		px[0] = Yl;
		px[1] = Cb;
		px[2] = Cr;
	}
}

int orc_encode(const uint8_t *body, int X, int Y, int Nf, int P, int H, int V, int q,
               int optimize, int restart_interval,
               orc_image **img_out, uint8_t **jpg, size_t *jpg_len)
{
	orc_image *img = image_new();
	float *frame = NULL, *plane = NULL;
	hufftab (*enc)[4] = calloc(2, sizeof(hufftab[4]));
	size_t (*freq)[4][257] = calloc(2, sizeof(size_t[4][257]));
	sink s = {0};
	int err = ORC_OK;

	*jpg = NULL;
	*jpg_len = 0;
	if (img_out) {
		*img_out = NULL;
	}
	if (img == NULL || enc == NULL || freq == NULL) {
		err = ORC_ERR_MEMORY;
		goto out;
	}
	if ((Nf != 1 && Nf != 3) || X <= 0 || Y <= 0 || X > 65535 || Y > 65535 || P < 1 || P > 16) {
		err = ORC_ERR_UNSUPPORTED;
		goto out;
	}

	/* encoder.c:104-152 read_image: component set-up, quantisation tables */
	img->P = P; img->X = X; img->Y = Y; img->Nf = Nf;
	img->ncomp = Nf;
	if (Nf == 1) {
		img->comp[0] = (orc_comp){1, 1, 1, 0, 0, 0, 0, 0, NULL};
		img->max_H = img->max_V = 1;
	} else {
		if (H < 1 || H > 2 || V < 1 || V > 2) {
			err = ORC_ERR_UNSUPPORTED;   /* assert in the reference (encoder.c:123-124) */
			goto out;
		}
		img->comp[0] = (orc_comp){1, H, V, 0, 0, 0, 0, 0, NULL};
		img->comp[1] = (orc_comp){2, 1, 1, 1, 1, 1, 0, 0, NULL};
		img->comp[2] = (orc_comp){3, 1, 1, 1, 1, 1, 0, 0, NULL};
		img->max_H = H;
		img->max_V = V;
	}
	orc_set_qtable(img->Q[0], 0, q);
	orc_set_qtable(img->Q[1], 1, q);
	err = image_alloc(img);
	if (err) {
		goto out;
	}

	size_t sx = img->size_x, sy = img->size_y, nf = (size_t)Nf;
	frame = malloc(sizeof(float) * nf * sx * sy);
	if (frame == NULL) {
		err = ORC_ERR_MEMORY;
		goto out;
	}

	/* frame.c:277-350 read_frame_body: samples -> float, replicate last column / row */
	{
		int maxval = (1 << P) - 1;
		size_t ss = maxval <= 255 ? 1 : 2;
		const uint8_t *r = body;
		for (size_t y = 0; y < (size_t)Y; ++y) {
			for (size_t x = 0; x < (size_t)X; ++x) {
				for (size_t c = 0; c < nf; ++c) {
					unsigned v = ss == 1 ? r[0] : (unsigned)((r[0] << 8) | r[1]);
					r += ss;
					frame[y * sx * nf + x * nf + c] = (float)v;
				}
			}
			for (size_t x = (size_t)X; x < sx; ++x) {
				for (size_t c = 0; c < nf; ++c) {
					frame[y * sx * nf + x * nf + c] = frame[y * sx * nf + ((size_t)X - 1) * nf + c];
				}
			}
		}
		for (size_t y = (size_t)Y; y < sy; ++y) {
			for (size_t x = 0; x < sx * nf; ++x) {
				frame[y * sx * nf + x] = frame[((size_t)Y - 1) * sx * nf + x];
			}
		}
	}

	/* frame.c:154-186 frame_to_ycc: only x < X, y < Y (padding keeps raw RGB, quirk Q7) */
	if (Nf == 3) {
		for (size_t y = 0; y < (size_t)Y; ++y) {
			orc_frame_to_ycc(&frame[y * sx * 3], (size_t)X, P);
		}
	}

	/* frame.c:84-132 box-mean downsample, imgproc.c:264-293 raster->blocks,
	 * imgproc.c:202-230 level shift + FDCT, imgproc.c:58-82 quantise */
	lut_init();
	for (int k = 0; k < img->ncomp; ++k) {
		orc_comp *c = &img->comp[k];
		size_t cx = c->bx * 8, cy = c->by * 8;
		size_t stx = sx / cx, sty = sy / cy;
		int shift = 1 << (P - 1);
		plane = malloc(sizeof(float) * cx * cy);
		if (plane == NULL) {
			err = ORC_ERR_MEMORY;
			goto out;
		}
		for (size_t y = 0; y < cy; ++y) {
			for (size_t x = 0; x < cx; ++x) {
				float px = 0.f;
				for (size_t yy = 0; yy < sty; ++yy) {
					for (size_t xx = 0; xx < stx; ++xx) {
						px += frame[(sty * y + yy) * sx * nf + nf * (stx * x + xx) + (size_t)k];
					}
				}
				px /= sty * stx;
				plane[y * cx + x] = px;
			}
		}
		for (size_t by = 0; by < c->by; ++by) {
			for (size_t bx = 0; bx < c->bx; ++bx) {
				float f[64];
				for (int v = 0; v < 8; ++v) {
					for (int u = 0; u < 8; ++u) {
						f[v * 8 + u] = plane[(by * 8 + (size_t)v) * cx + bx * 8 + (size_t)u];
					}
				}
				for (int j = 0; j < 64; ++j) {
					f[j] -= shift;
				}
				orc_fdct_block(f);
				orc_quantize_block(&c->coef[(by * c->bx + bx) * 64], f, img->Q[c->Tq]);
			}
		}
		img->plane[k] = plane;      /* kept for the stage tests (tests/: float stage boundaries) */
		plane = NULL;
	}

	/* encoder.c:589-645 produce_codestream */
	sink_marker(&s, 0xd8);
	for (int t = 0; t < (Nf > 1 ? 2 : 1); ++t) {                 /* DQT, always 8-bit entries */
		sink_marker(&s, 0xdb);
		sink_word(&s, 67);
		sink_put(&s, (uint8_t)t);
		for (int i = 0; i < 64; ++i) {
			sink_put(&s, (uint8_t)img->Q[t][zz[i]]);
		}
	}
	sink_marker(&s, 0xc0);                                       /* SOF0 */
	sink_word(&s, 8u + 3u * (unsigned)Nf);
	sink_put(&s, (uint8_t)P);
	sink_word(&s, (unsigned)Y);
	sink_word(&s, (unsigned)X);
	sink_put(&s, (uint8_t)Nf);
	for (int k = 0; k < img->ncomp; ++k) {
		sink_put(&s, (uint8_t)img->comp[k].id);
		sink_put(&s, (uint8_t)((img->comp[k].H << 4) | img->comp[k].V));
		sink_put(&s, (uint8_t)img->comp[k].Tq);
	}

	int ntab = Nf > 1 ? 2 : 1;
	if (optimize) {                                              /* encoder.c:525-558 */
		for (int tc = 0; tc < 2; ++tc) {
			for (int th = 0; th < 4; ++th) {
				freq[tc][th][256] = 1;                           /* common.c:219 */
			}
		}
		err = walk_mcus(img, NULL, enc, freq, restart_interval);
		if (err) {
			goto out;
		}
		for (int tc = 0; tc < 2; ++tc) {
			for (int th = 0; th < ntab; ++th) {
				orc_adapt_table(freq[tc][th], img->htL[tc][th], img->htV[tc][th]);
			}
		}
	}
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 2; ++th) {
			hufftab_build(&enc[tc][th], img->htL[tc][th], img->htV[tc][th]);
			/* value_to_vlc(): first matching entry wins (huffman.c:238-245) */
			orc_make_codes(img->htL[tc][th], img->htV[tc][th], enc[tc][th].ehufsi, enc[tc][th].ehufco);
		}
	}
	for (int th = 0; th < ntab; ++th) {                          /* DHT: DC then AC per table id */
		put_dht(&s, 0, th, img->htL[0][th], img->htV[0][th]);
		put_dht(&s, 1, th, img->htL[1][th], img->htV[1][th]);
	}
	if (restart_interval > 0) {                                  /* extension: DRI */
		sink_marker(&s, 0xdd);
		sink_word(&s, 4);
		sink_word(&s, (unsigned)restart_interval);
		img->Ri = restart_interval;
	}
	sink_marker(&s, 0xda);                                       /* SOS (encoder.c:345-400) */
	sink_word(&s, 6u + 2u * (unsigned)Nf);
	sink_put(&s, (uint8_t)Nf);
	for (int k = 0; k < img->ncomp; ++k) {
		sink_put(&s, (uint8_t)img->comp[k].id);
		sink_put(&s, (uint8_t)((img->comp[k].Td << 4) | img->comp[k].Ta));
	}
	sink_put(&s, 0);
	sink_put(&s, 63);
	sink_put(&s, 0);
	err = walk_mcus(img, &s, enc, NULL, restart_interval);
	if (err) {
		goto out;
	}
	sink_marker(&s, 0xd9);
	if (s.oom) {
		err = ORC_ERR_MEMORY;
		goto out;
	}
	*jpg = s.p;
	*jpg_len = s.n;
	s.p = NULL;
out:
	free(s.p);
	free(frame);
	free(plane);
	free(enc);
	free(freq);
	if (err == ORC_OK && img_out) {
		*img_out = img;
	} else {
		orc_image_free(img);
	}
	return err;
}

/*
 * EXTENSION (test input generator; the reference encoder writes one interleaved scan only):
 * the image of orc_encode() coded in SEVERAL SCANS, which the reference DEcoder reads scan by scan
 * (decoder.c:593-601): `split` 0 = one scan per component, 1 = {first} + {the others}, 2 = {all but the
 * last} + {last}.  A scan of one component follows the reference's Ns == 1 walk (decoder.c:274-302: the
 * blocks in raster order over the padded grid b_x x b_y, quirk Q8; restart_interval counts blocks there);
 * a scan of several follows the MCU walk restricted to its components (decoder.c:303-358; the interval
 * counts MCUs).  DC predictors start at 0 in every scan.  redefine != 0: every scan names table ids 0/0
 * and a DHT in front of it redefines them (tables "as they stand when the scan starts",
 * decoder.c:581-591) -- one-component scans only.
 * The judge of these streams is the reference decoder (tests/golden/make_golden.py).
 */
typedef struct {
	int n;
	int comp[ORC_MAXC];
} scan_set;

/* one scan: statistics (freq != NULL) or bits */
static int scans_walk(orc_image *img, const scan_set *sc, int ri, sink *s, hufftab *dc, hufftab *ac,
                      size_t (*freq)[4][257], int td_override, int ta_override)
{
	bitw w = {s, 0, 0};
	int32_t pred[ORC_MAXC] = {0};
	unsigned rst = 0;
	size_t units = sc->n == 1 ? img->comp[sc->comp[0]].bx * img->comp[sc->comp[0]].by : img->m_x * img->m_y;

	for (size_t u = 0; u < units; ++u) {
		if (ri > 0 && u > 0 && u % (size_t)ri == 0) {
			if (!freq) {
				bitw_flush(&w);
				sink_marker(s, 0xd0 + (rst & 7));
			}
			rst++;
			memset(pred, 0, sizeof pred);
		}
		for (int j = 0; j < sc->n; ++j) {
			orc_comp *c = &img->comp[sc->comp[j]];
			int nv = sc->n == 1 ? 1 : c->V, nh = sc->n == 1 ? 1 : c->H;
			for (int v = 0; v < nv; ++v) {
				for (int h = 0; h < nh; ++h) {
					size_t b = u;
					if (sc->n > 1) {
						size_t x = u % img->m_x, y = u / img->m_x;
						b = (y * (size_t)c->V + (size_t)v) * c->bx + x * (size_t)c->H + (size_t)h;
					}
					int32_t blk[64];
					memcpy(blk, &c->coef[b * 64], sizeof blk);
					int32_t dcv = blk[0];
					blk[0] -= pred[j];
					pred[j] = dcv;
					if (blk[0] < -2047 || blk[0] > 2047) {
						return ORC_ERR_DC_RANGE;
					}
					int td = td_override >= 0 ? td_override : c->Td, ta = ta_override >= 0 ? ta_override : c->Ta;
					int err = freq ? code_block(NULL, NULL, NULL, freq[0][c->Td], freq[1][c->Ta], blk)
					               : code_block(&w, &dc[td], &ac[ta], NULL, NULL, blk);
					if (err) {
						return err;
					}
				}
			}
		}
	}
	if (!freq) {
		bitw_flush(&w);
	}
	return ORC_OK;
}

int orc_encode_scans(const uint8_t *body, int X, int Y, int Nf, int P, int H, int V, int q,
                     int optimize, int restart_interval, int redefine, int split,
                     uint8_t **jpg, size_t *jpg_len)
{
	orc_image *img = NULL;
	uint8_t *first = NULL;
	size_t first_len = 0;
	sink s = {0};
	hufftab *dc = NULL, *ac = NULL;
	scan_set scans[ORC_MAXC];
	int nscans = 0;

	*jpg = NULL;
	*jpg_len = 0;
	int err = orc_encode(body, X, Y, Nf, P, H, V, q, optimize, 0, &img, &first, &first_len);
	if (err) {
		return err;
	}
	free(first);
	if (split == 0 || img->ncomp < 3) {
		for (int k = 0; k < img->ncomp; ++k) {
			scans[nscans].n = 1;
			scans[nscans++].comp[0] = k;
		}
	} else {
		int cut = split == 1 ? 1 : img->ncomp - 1;
		scans[0].n = cut;
		scans[1].n = img->ncomp - cut;
		for (int k = 0; k < img->ncomp; ++k) {
			if (k < cut) {
				scans[0].comp[k] = k;
			} else {
				scans[1].comp[k - cut] = k;
			}
		}
		nscans = 2;
		redefine = 0;
	}
	dc = calloc(4, sizeof *dc);
	ac = calloc(4, sizeof *ac);
	if (dc == NULL || ac == NULL) {
		err = ORC_ERR_MEMORY;
		goto out;
	}
	if (optimize) {
		/* the DC differences follow the order of THESE walks: statistics and tables again (Annex K.2) */
		size_t (*freq)[4][257] = calloc(2, sizeof(size_t[4][257]));
		if (freq == NULL) {
			err = ORC_ERR_MEMORY;
			goto out;
		}
		for (int tc = 0; tc < 2; ++tc) {
			for (int th = 0; th < 4; ++th) {
				freq[tc][th][256] = 1;
			}
		}
		for (int i = 0; i < nscans && !err; ++i) {
			err = scans_walk(img, &scans[i], restart_interval, NULL, NULL, NULL, freq, -1, -1);
		}
		if (!err) {
			for (int tc = 0; tc < 2; ++tc) {
				for (int th = 0; th < (Nf > 1 ? 2 : 1); ++th) {
					orc_adapt_table(freq[tc][th], img->htL[tc][th], img->htV[tc][th]);
				}
			}
		}
		free(freq);
		if (err) {
			goto out;
		}
	}
	for (int th = 0; th < 2; ++th) {
		hufftab_build(&dc[th], img->htL[0][th], img->htV[0][th]);
		orc_make_codes(img->htL[0][th], img->htV[0][th], dc[th].ehufsi, dc[th].ehufco);
		hufftab_build(&ac[th], img->htL[1][th], img->htV[1][th]);
		orc_make_codes(img->htL[1][th], img->htV[1][th], ac[th].ehufsi, ac[th].ehufco);
	}
	sink_marker(&s, 0xd8);
	for (int t = 0; t < (Nf > 1 ? 2 : 1); ++t) {
		sink_marker(&s, 0xdb);
		sink_word(&s, 67);
		sink_put(&s, (uint8_t)t);
		for (int i = 0; i < 64; ++i) {
			sink_put(&s, (uint8_t)img->Q[t][zz[i]]);
		}
	}
	sink_marker(&s, 0xc0);
	sink_word(&s, 8u + 3u * (unsigned)Nf);
	sink_put(&s, (uint8_t)P);
	sink_word(&s, (unsigned)Y);
	sink_word(&s, (unsigned)X);
	sink_put(&s, (uint8_t)Nf);
	for (int k = 0; k < img->ncomp; ++k) {
		sink_put(&s, (uint8_t)img->comp[k].id);
		sink_put(&s, (uint8_t)((img->comp[k].H << 4) | img->comp[k].V));
		sink_put(&s, (uint8_t)img->comp[k].Tq);
	}
	if (!redefine) {
		for (int th = 0; th < (Nf > 1 ? 2 : 1); ++th) {
			put_dht(&s, 0, th, img->htL[0][th], img->htV[0][th]);
			put_dht(&s, 1, th, img->htL[1][th], img->htV[1][th]);
		}
	}
	if (restart_interval > 0) {
		sink_marker(&s, 0xdd);
		sink_word(&s, 4);
		sink_word(&s, (unsigned)restart_interval);
	}
	for (int i = 0; i < nscans; ++i) {
		const scan_set *sc = &scans[i];
		int tdo = -1, tao = -1;
		if (redefine) {                                          /* one component: its tables under ids 0/0 */
			orc_comp *c = &img->comp[sc->comp[0]];
			put_dht(&s, 0, 0, img->htL[0][c->Td], img->htV[0][c->Td]);
			put_dht(&s, 1, 0, img->htL[1][c->Ta], img->htV[1][c->Ta]);
			dc[2] = dc[c->Td];
			ac[2] = ac[c->Ta];
			tdo = tao = 2;                                       /* the codes of the table that now sits under id 0 */
		}
		sink_marker(&s, 0xda);
		sink_word(&s, 6u + 2u * (unsigned)sc->n);
		sink_put(&s, (uint8_t)sc->n);
		for (int j = 0; j < sc->n; ++j) {
			orc_comp *c = &img->comp[sc->comp[j]];
			sink_put(&s, (uint8_t)c->id);
			sink_put(&s, (uint8_t)(redefine ? 0 : ((c->Td << 4) | c->Ta)));
		}
		sink_put(&s, 0);
		sink_put(&s, 63);
		sink_put(&s, 0);
		err = scans_walk(img, sc, restart_interval, &s, dc, ac, NULL, tdo, tao);
		if (err) {
			goto out;
		}
	}
	sink_marker(&s, 0xd9);
	if (s.oom) {
		err = ORC_ERR_MEMORY;
		goto out;
	}
	*jpg = s.p;
	*jpg_len = s.n;
	s.p = NULL;
out:
	free(s.p);
	free(dc);
	free(ac);
	orc_image_free(img);
	return err;
}

/* ======================================================================== accessors */

void orc_img_info(const orc_image *img, int *P, int *X, int *Y, int *Nf, int *ncomp,
                  size_t *m_x, size_t *m_y, size_t *size_x, size_t *size_y, size_t *mblocks)
{
	*P = img->P; *X = img->X; *Y = img->Y; *Nf = img->Nf; *ncomp = img->ncomp;
	*m_x = img->m_x; *m_y = img->m_y; *size_x = img->size_x; *size_y = img->size_y;
	*mblocks = img->mblocks;
}

void orc_img_comp(const orc_image *img, int k, int *id, int *H, int *V, int *Tq, int *Td, int *Ta,
                  size_t *bx, size_t *by)
{
	const orc_comp *c = &img->comp[k];
	*id = c->id; *H = c->H; *V = c->V; *Tq = c->Tq; *Td = c->Td; *Ta = c->Ta;
	*bx = c->bx; *by = c->by;
}

const int32_t *orc_img_coef(const orc_image *img, int k) { return img->comp[k].coef; }
const uint16_t *orc_img_qtable(const orc_image *img, int t) { return img->Q[t & 3]; }
const float *orc_img_frame(const orc_image *img) { return img->frame; }
const float *orc_img_plane(const orc_image *img, int k) { return k >= 0 && k < ORC_MAXC ? img->plane[k] : NULL; }

void orc_img_htable(const orc_image *img, int Tc, int Th, uint8_t L[16], uint8_t V[256])
{
	memcpy(L, img->htL[Tc & 1][Th & 3], 16);
	memcpy(V, img->htV[Tc & 1][Th & 3], 256);
}
