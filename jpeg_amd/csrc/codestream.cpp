// jpeg_amd/csrc/codestream.cpp -- host-side codestream layer.
//
// In-memory restatement of the serial, byte-sized parts of the reference: marker walk and segment
// parsing (decoder.c:20-259, 472-659; io.c:186-220), Annex C table generation (huffman.c:45-178),
// the Annex K.2 optimiser (huffman.c:330-537), quality scaling (encoder.c:38-65) and header
// production (encoder.c:195-400).  None of this is worth a kernel (<= a few KB per frame).
#include <stddef.h>
#include <string.h>
#include "../../include/jpeg_amd.h"
#include "codestream.hpp"

// ---- constant tables ---------------------------------------------------------------------------

// Annex K.1 (encoder.c:14-34), raster order
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

// Annex K.3 = the reference's implicit Motion-JPEG tables (mjpeg.h:5-82): BITS then HUFFVAL
static const uint8_t k3_bits[2][2][16] = {
	{{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}},
	{{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}}
};
static const uint8_t k3_ac_vals[2][162] = {
	{
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
	},
	{
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
	}
};

void jh_default_table(int Tc, int Th, JhHuffSpec *s)
{
	memset(s, 0, sizeof *s);
	memcpy(s->L, k3_bits[Tc & 1][Th & 1], 16);
	if (Tc == 0) {
		for (int i = 0; i < 12; ++i) {
			s->V[i] = (uint8_t)i;
		}
	} else {
		memcpy(s->V, k3_ac_vals[Th & 1], 162);
	}
}

void jh_set_qtable(uint16_t Q[64], int chroma, int q)
{
	const uint8_t *base = chroma ? k1_chroma : k1_luma;
	q = q < 1 ? 1 : (q > 100 ? 100 : q);                       // encoder.c:40-45
	const int sf = q < 50 ? 5000 / q : 200 - 2 * q;            // encoder.c:49-53
	for (int i = 0; i < 64; ++i) {
		int v = (int)((base[i] * (unsigned)sf + 50u) / 100u);  // encoder.c:63
		Q[i] = (uint16_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
	}
}

// ---- Annex C ------------------------------------------------------------------------------------

struct CanonCodes {
	int n;
	uint8_t size[257];
	uint16_t code[257];
};

static int canon_codes(const JhHuffSpec *s, CanonCodes *c)
{
	int k = 0;
	for (int len = 1; len <= 16; ++len) {                      // Figure C.1
		for (int j = 0; j < s->L[len - 1]; ++j) {
			if (k >= 256) {
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			c->size[k++] = (uint8_t)len;
		}
	}
	c->size[k] = 0;
	c->n = k;
	unsigned code = 0;                                         // Figure C.2
	int si = c->size[0];
	for (int p = 0; p < k;) {
		while (p < k && c->size[p] == si) {
			c->code[p++] = (uint16_t)code++;
		}
		code <<= 1;
		si++;
	}
	return RET_SUCCESS;
}

int jh_build_declut(const JhHuffSpec *s, int tc, JhHuffLut *lut)
{
	CanonCodes c;
	int err = canon_codes(s, &c);
	if (err) {
		return err;
	}
	static_assert(sizeof(JhHuffLut) % 16 == 0, "table offsets are kept in units of 16 bytes");
	static_assert(offsetof(JhHuffLut, sub) == JH_LUT_DC_BYTES && JH_LUT_DC_BYTES % 16 == 0, "short form of a DC table");
	static_assert(offsetof(JhHuffLut, pair) == JH_LUT_PAIR_OFF, "pair[] is read at a fixed distance from fast[]");
	memset(lut, 0, sizeof *lut);
	memcpy(lut->huffval, s->V, 256);
	// query_code() (huffman.c:193-225) returns the FIRST table entry that matches; canonical
	// codes are prefix-free, so filling in table order without overwriting reproduces that.
	int nsub = 0;
	for (int p = 0; p < c.n; ++p) {
		const int len = c.size[p];
		// (code length + category, zero run, category); EOB of an AC table = run 63 (huffcore.hpp)
		const unsigned sym = s->V[p];
		const unsigned cat = sym & 15u;
		const unsigned run = (tc == 1 && sym == 0) ? 63u : (sym >> 4);
		const uint16_t entry = (uint16_t)(((len + cat) << 10) | (run << 4) | cat);
		if (len <= JH_LUT_BITS) {
			// every JH_LUT_BITS-bit window that starts with this code
			const unsigned first = (unsigned)c.code[p] << (JH_LUT_BITS - len);
			const unsigned count = 1u << (JH_LUT_BITS - len);
			for (unsigned i = 0; i < count && first + i < (1u << JH_LUT_BITS); ++i) {
				if (lut->fast[first + i] == 0) {
					lut->fast[first + i] = entry;
				}
			}
			continue;
		}
		// longer code: second-level table of its JH_LUT_BITS-bit prefix, indexed by the next 6 bits
		const unsigned prefix = ((unsigned)c.code[p] >> (len - JH_LUT_BITS)) & ((1u << JH_LUT_BITS) - 1u);
		uint16_t &f = lut->fast[prefix];
		if (f == 0) {
			// (DC tables: no second-level tables -- the kernels' copy of a DC table ends in front of them)
			f = tc == 1 && nsub < JH_SUB_TABLES ? (uint16_t)(0x8000u | (unsigned)nsub++) : (uint16_t)JH_LUT_SEARCH;
		}
		if (!(f & 0x8000u)) {
			continue;                                          // prefix taken by a shorter code (malformed table)
		}
		if (f == JH_LUT_SEARCH) {
			continue;
		}
		uint16_t *sub = lut->sub[f & 0x7fffu];
		const unsigned first = ((unsigned)c.code[p] << (16 - len)) & 63u;
		const unsigned count = 1u << (16 - len);
		for (unsigned i = 0; i < count && first + i < 64u; ++i) {
			if (sub[first + i] == 0) {
				sub[first + i] = entry;
			}
		}
	}
	for (unsigned i = 0; i < (1u << JH_LUT_BITS); ++i) {
		if (lut->fast[i] == 0) {
			lut->fast[i] = (uint16_t)JH_LUT_INVALID;           // no code starts with these bits
		}
	}
	// AC tables: the symbol BEHIND the window's first one, where its code lies inside the window too (common.hpp).
	// The bits of the window behind the first symbol (its magnitude bits included) index the table once more, the
	// unknown bits behind the window as zeros: the entry found there is the second symbol iff its code is no longer
	// than the bits that are known.
	if (tc == 1) {
		for (unsigned i = 0; i < (1u << JH_LUT_BITS); ++i) {
			const uint32_t e1 = lut->fast[i];
			const unsigned total1 = e1 >> 10, run1 = (e1 >> 4) & 63u;
			if ((e1 & 0x8000u) || run1 == 63u || total1 >= JH_LUT_BITS) {
				continue;                                      // long code / none, EOB, or nothing of the window left
			}
			const unsigned known = JH_LUT_BITS - total1;
			const uint32_t e2 = lut->fast[(i << total1) & ((1u << JH_LUT_BITS) - 1u)];
			if (e2 & 0x8000u) {
				continue;
			}
			const unsigned total2 = e2 >> 10, run2 = (e2 >> 4) & 63u, cat2 = e2 & 15u;
			if (total2 - cat2 > known || total1 + total2 > 31u) {
				continue;                                      // the code reaches past the window
			}
			lut->pair[i] = (uint16_t)(((total1 + total2) << 11) | ((run1 + 1u + run2) << 4) | cat2);
		}
	}
	// slow path: left-aligned upper limits per length (canonical MAXCODE of F.2.2.3)
	uint32_t lim = 0;
	for (int len = 1, p = 0; len <= 16; ++len) {
		const int cnt = s->L[len - 1];
		if (cnt) {
			lut->offs[len] = p - (int)c.code[p];
			lim = ((uint32_t)c.code[p + cnt - 1] + 1u) << (16 - len);
			p += cnt;
		}
		lut->limit[len] = lim;
	}
	lut->limit[0] = 0;
	return RET_SUCCESS;
}

int jh_build_enccodes(const JhHuffSpec *s, JhHuffEnc *e)
{
	CanonCodes c;
	int err = canon_codes(s, &c);
	if (err) {
		return err;
	}
	memset(e, 0, sizeof *e);
	for (int p = c.n - 1; p >= 0; --p) {                       // first match wins (huffman.c:238-245)
		e->code[s->V[p]] = ((uint32_t)c.code[p] << 8) | c.size[p];
	}
	return RET_SUCCESS;
}

// ---- geometry -----------------------------------------------------------------------------------

int jh_finish_geometry(JhGeom *g)
{
	if (g->ncomp < 1 || g->ncomp > JH_MAX_COMP || g->X <= 0 || g->Y <= 0 || g->P < 2 || g->P > 16) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	if (g->ncomp == 1) {
		// Ns == 1 walk: plain raster over the padded block grid (decoder.c:285-289); same
		// result as H = V = 1 over that grid.
		const int H = g->H[0], V = g->V[0];
		if (H < 1 || H > 4 || V < 1 || V > 4) {
			return RET_FAILURE_FILE_UNSUPPORTED;               // the reference divides by zero (common.c:171-172)
		}
		const int m_x = (g->X + 8 * H - 1) / (8 * H), m_y = (g->Y + 8 * V - 1) / (8 * V);
		g->H[0] = g->V[0] = 1;
		g->max_H = g->max_V = 1;
		g->m_x = m_x * H;
		g->m_y = m_y * V;
	} else {
		g->max_H = g->max_V = 0;
		for (int k = 0; k < g->ncomp; ++k) {
			if (g->H[k] < 1 || g->H[k] > 4 || g->V[k] < 1 || g->V[k] > 4) {
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			g->max_H = g->H[k] > g->max_H ? g->H[k] : g->max_H;
			g->max_V = g->V[k] > g->max_V ? g->V[k] : g->max_V;
		}
		g->m_x = (g->X + 8 * g->max_H - 1) / (8 * g->max_H);   // common.c:171-172
		g->m_y = (g->Y + 8 * g->max_V - 1) / (8 * g->max_V);
	}
	g->size_x = g->m_x * 8 * g->max_H;                         // frame.c:21-22
	g->size_y = g->m_y * 8 * g->max_V;
	g->B = 0;
	// blocks of an MCU follow the order of the scan header (decoder.c:319-357), components
	// themselves stay in ascending-id order (that is the order of the output frame, frame.c:49-80)
	unsigned seen = 0;
	for (int j = 0; j < g->ncomp; ++j) {
		const int k = g->scan_order[j];
		if (k < 0 || k >= g->ncomp || (seen & (1u << k))) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		seen |= 1u << k;
		g->bx[k] = g->m_x * g->H[k];                           // common.c:181-182
		g->by[k] = g->m_y * g->V[k];
		if (g->max_H % g->H[k] || g->max_V % g->V[k]) {
			return RET_FAILURE_FILE_UNSUPPORTED;               // frame.c:57-58 would leave holes
		}
		g->step_x[k] = g->max_H / g->H[k];
		g->step_y[k] = g->max_V / g->V[k];
		g->comp_first[k] = g->B;
		for (int v = 0; v < g->V[k]; ++v) {
			for (int h = 0; h < g->H[k]; ++h) {
				if (g->B >= JH_MAX_MCU_BLOCKS) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				g->blk_comp[g->B] = k;
				g->blk_h[g->B] = h;
				g->blk_v[g->B] = v;
				g->B++;
			}
		}
	}
	g->nmcu = g->m_x * g->m_y;
	if ((long long)g->nmcu * g->B > 0x3fffffff) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	g->nblk = g->nmcu * g->B;
	g->out_comps = g->Nf == 1 ? 1 : 3;                         // frame.c:554-561
	g->sample_bytes = g->P <= 8 ? 1 : 2;                       // frame.c:246-257
	return RET_SUCCESS;
}

// ---- marker parser ---------------------------------------------------------------------------------

namespace {
struct Reader {
	const uint8_t *p;
	size_t n, pos;
	bool ok(size_t k) const { return pos + k <= n; }
	unsigned u8() { return p[pos++]; }
	unsigned u16() { unsigned v = (p[pos] << 8) | p[pos + 1]; pos += 2; return v; }
};

// io.c:186-220 read_marker: skip to FF, swallow fill bytes, FF00 restarts the search
int next_marker(Reader &r, unsigned *m)
{
	for (;;) {
		do {
			if (!r.ok(1)) {
				return RET_FAILURE_FILE_IO;
			}
		} while (r.u8() != 0xff);
		for (;;) {
			if (!r.ok(1)) {
				return RET_FAILURE_FILE_IO;
			}
			unsigned b = r.u8();
			if (b == 0xff) {
				continue;
			}
			if (b == 0) {
				break;
			}
			*m = 0xff00u | b;
			return RET_SUCCESS;
		}
	}
}
}  // namespace

// Motion JPEG: a stream of independent SOI..EOI codestreams (the reference's process_jpeg_stream()
// handles exactly one and reports the rest as trailing garbage, decoder.c:603-612).  Segments with a
// length field are skipped by length (an FF D9 inside an APPn payload is not an EOI); entropy-coded
// data is scanned for the first marker that is neither a stuffed FF00 nor RSTn (io.c:247-274).
int jh_split_stream(const uint8_t *p, size_t n, size_t *off, size_t *len, int cap)
{
	int count = 0;
	size_t i = 0;
	while (i + 1 < n) {
		if (!(p[i] == 0xff && p[i + 1] == 0xd8)) {
			++i;                                               // garbage between frames
			continue;
		}
		const size_t start = i;
		i += 2;
		bool closed = false;
		while (i + 1 < n && !closed) {
			if (p[i] != 0xff) {
				++i;
				continue;
			}
			const unsigned m = p[i + 1];
			if (m == 0xff) {
				++i;                                           // fill byte
				continue;
			}
			if (m == 0x00 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) {
				i += 2;                                        // stuffed FF, TEM, RSTn: no payload
				continue;
			}
			if (m == 0xd9) {
				i += 2;
				closed = true;
				break;
			}
			if (m == 0xd8) {
				break;                                         // a new SOI before any EOI: the frame was cut short
			}
			if (i + 3 >= n) {
				i = n;
				break;
			}
			const size_t seglen = ((size_t)p[i + 2] << 8) | p[i + 3];
			i += 2 + (seglen < 2 ? 2 : seglen);                // after SOS this lands on the entropy-coded data
		}
		if (i > n) {
			i = n;
		}
		if (!closed && !(i + 1 < n)) {
			i = n;                                             // no EOI: the last frame runs to the end of the data
		}
		if (count < cap) {
			off[count] = start;
			len[count] = i - start;
		}
		++count;
	}
	return count;
}

// parse_format (decoder.c:472-659) from r.pos up to and including the next SOS header.  eoi != NULL: an EOI
// ends the walk with *eoi = true (the caller is between scans); eoi == NULL: an EOI before the first scan
// is an error.
static int parse_segments(Reader &r, JhHeader *h, bool *eoi)
{
	const uint8_t *p = r.p;
	const size_t n = r.n;
	bool &have_sof = h->have_sof;

	for (;;) {
		unsigned m;
		int err = next_marker(r, &m);
		if (err) {
			return err;
		}
		if (m == 0xffd8 || m == 0xff01) {                      // SOI, TEM
			continue;
		}
		// The reference knows APP0-8 and APP11-14 only: APP9, APP10, APP15, DNL and the JPGn range end in
		// "unhandled marker" (decoder.c:654-656).  That is kept, bug for bug, unless the caller asks for
		// tolerance: then these segments are skipped by their length like the others.
		const bool extra = h->tolerant && (m == 0xffe9 || m == 0xffea || m == 0xffef || m == 0xffdc ||
		                                   (m >= 0xfff0 && m <= 0xfffd));
		if ((m >= 0xffe0 && m <= 0xffe8) || (m >= 0xffeb && m <= 0xffee) || m == 0xffcc || m == 0xfffe || extra) {
			// APPn except 9, 10, 15 (decoder.c:498-516), DAC (:647-653), COM (:635-641)
			if (!r.ok(2)) {
				return RET_FAILURE_FILE_IO;
			}
			size_t len = ((size_t)p[r.pos] << 8) | p[r.pos + 1];
			if (m == 0xfffe && len < 2) {
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			r.pos += len;
			if (r.pos > n) {
				return RET_FAILURE_FILE_IO;
			}
			continue;
		}
		if (m == 0xffdb) {                                     // DQT, decoder.c:20-68 + :518-527
			const size_t seg = r.pos;
			if (!r.ok(2)) {
				return RET_FAILURE_FILE_IO;
			}
			const size_t len = r.u16();
			do {
				if (!r.ok(1)) {
					return RET_FAILURE_FILE_IO;
				}
				unsigned b = r.u8();
				const int Pq = b >> 4, Tq = b & 15;
				if (Tq >= 4 || Pq >= 2) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				if (!r.ok(Pq ? 128 : 64)) {
					return RET_FAILURE_FILE_IO;
				}
				for (int i = 0; i < 64; ++i) {
					h->Q[Tq][jh_zigzag[i]] = (uint16_t)(Pq ? r.u16() : r.u8());
				}
			} while (r.pos < seg + len);
			continue;
		}
		if (m == 0xffc0 || m == 0xffc1) {                      // SOF0 / SOF1, decoder.c:70-136
			if (have_sof) {
				// A second frame header.  The reference allocates new buffers and goes on (decoder.c:529-543);
				// scans already described here keep the component indices and tables of the first one, so a
				// change of the frame under them is refused instead of reinterpreted.
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			if (!r.ok(8)) {
				return RET_FAILURE_FILE_IO;
			}
			r.pos += 2;
			JhGeom &g = h->g;
			g.P = (int)r.u8();
			g.Y = (int)r.u16();
			g.X = (int)r.u16();
			g.Nf = (int)r.u8();
			if (g.Nf < 1 || g.Nf > JH_MAX_COMP || g.Nf == 2) {
				return RET_FAILURE_FILE_UNSUPPORTED;           // frame_to_rgb abort()s on 2 (frame.c:239)
			}
			if (!r.ok(3 * (size_t)g.Nf)) {
				return RET_FAILURE_FILE_IO;
			}
			g.ncomp = 0;
			for (int i = 0; i < g.Nf; ++i) {
				const int id = (int)r.u8();
				const unsigned hv = r.u8();
				const int Tq = (int)r.u8();
				int k = g.ncomp;
				for (int j = 0; j < g.ncomp; ++j) {
					if (g.id[j] == id) {
						return RET_FAILURE_FILE_UNSUPPORTED;   // duplicate ids: not on the GPU path
					}
				}
				while (k > 0 && g.id[k - 1] > id) {            // ascending id (the reference walks 0..255)
					g.id[k] = g.id[k - 1]; g.H[k] = g.H[k - 1]; g.V[k] = g.V[k - 1]; g.Tq[k] = g.Tq[k - 1];
					k--;
				}
				g.id[k] = id;
				g.H[k] = hv >> 4;
				g.V[k] = hv & 15;
				g.Tq[k] = Tq;
				if (Tq >= 4) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				g.ncomp++;
			}
			have_sof = true;
			continue;
		}
		if (m == 0xffc2 || m == 0xffc3 || m == 0xffc9 || m == 0xffca) {
			return RET_FAILURE_FILE_UNSUPPORTED;               // decoder.c:545-579
		}
		if (m == 0xffc4) {                                     // DHT, decoder.c:143-184 + :581-591
			const size_t seg = r.pos;
			if (!r.ok(2)) {
				return RET_FAILURE_FILE_IO;
			}
			const size_t len = r.u16();
			do {
				if (!r.ok(17)) {
					return RET_FAILURE_FILE_IO;
				}
				unsigned b = r.u8();
				const int Tc = b >> 4, Th = b & 15;
				if (Tc >= 2 || Th >= 4) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				JhHuffSpec &s = h->ht[Tc][Th];
				memset(&s, 0, sizeof s);
				size_t total = 0;
				for (int i = 0; i < 16; ++i) {
					s.L[i] = (uint8_t)r.u8();
					total += s.L[i];
				}
				if (total > 256) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				if (!r.ok(total)) {
					return RET_FAILURE_FILE_IO;
				}
				memcpy(s.V, p + r.pos, total);
				r.pos += total;
			} while (r.pos < seg + len);
			continue;
		}
		if (m == 0xffdd) {                                     // DRI, decoder.c:390-401 (stored, never used)
			if (!r.ok(4)) {
				return RET_FAILURE_FILE_IO;
			}
			r.pos += 2;
			h->g.Ri = (int)r.u16();
			continue;
		}
		if (m == 0xffda) {                                     // SOS, decoder.c:197-259
			if (!have_sof) {
				return RET_FAILURE_FILE_UNSUPPORTED;           // decoder.c:306-309
			}
			if (!r.ok(3)) {
				return RET_FAILURE_FILE_IO;
			}
			r.pos += 2;
			h->Ns = (int)r.u8();
			if (h->Ns < 1 || h->Ns > JH_MAX_COMP) {
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			if (!r.ok(2 * (size_t)h->Ns + 3)) {
				return RET_FAILURE_FILE_IO;                        // cut inside the scan header: read_byte() fails
			}
			for (int j = 0; j < h->Ns; ++j) {
				const int cs = (int)r.u8();
				const unsigned tt = r.u8();
				int k = -1;
				for (int i = 0; i < h->g.ncomp; ++i) {
					if (h->g.id[i] == cs) {
						k = i;
					}
				}
				if (k < 0 || (tt >> 4) >= 4 || (tt & 15) >= 4) {
					return RET_FAILURE_FILE_UNSUPPORTED;
				}
				h->scan_comp[j] = k;
				h->g.Td[k] = tt >> 4;
				h->g.Ta[k] = tt & 15;
			}
			const unsigned Ss = r.u8(), Se = r.u8(), AhAl = r.u8();
			if (Ss != 0 || Se != 63 || AhAl != 0) {
				return RET_FAILURE_FILE_UNSUPPORTED;           // decoder.c:240-250
			}
			h->ecs_offset = r.pos;
			return RET_SUCCESS;
		}
		if (m == 0xffd9) {
			if (eoi) {
				*eoi = true;
				return RET_SUCCESS;
			}
			if (have_sof) {
				// EOI before any scan: the reference's epilogue() makes an image of the zero coefficients
				// alloc_buffers() left (decoder.c:603-612, common.c:129).  Ns = 0 tells the caller.
				h->Ns = 0;
				h->ecs_offset = r.pos;
				return RET_SUCCESS;
			}
			return RET_FAILURE_FILE_UNSUPPORTED;               // and no frame header either: nothing to make an image of
		}
		return RET_FAILURE_FILE_UNSUPPORTED;                   // decoder.c:654-656
	}
}

int jh_parse_headers(const uint8_t *p, size_t n, JhHeader *h, bool tolerant)
{
	Reader r = {p, n, 0};
	memset(h, 0, sizeof *h);
	h->tolerant = tolerant;
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			h->Q[t][i] = 1;                                    // common.c:10-21
		}
	}
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 2; ++th) {
			jh_default_table(tc, th, &h->ht[tc][th]);          // common.c:90-94
		}
	}
	return parse_segments(r, h, nullptr);
}

int jh_parse_next_scan(const uint8_t *p, size_t n, size_t pos, JhHeader *h, bool *eoi)
{
	Reader r = {p, n, pos};
	*eoi = false;
	return parse_segments(r, h, eoi);
}

int jh_walk_to_restart(const uint8_t *p, size_t n, size_t pos, bool tolerant, size_t *rst)
{
	Reader r = {p, n, pos};
	*rst = 0;
	for (;;) {
		unsigned m;
		int err = next_marker(r, &m);
		if (err) {
			return err;
		}
		if (m == 0xffd8 || m == 0xff01) {                      // SOI, TEM
			continue;
		}
		if (m >= 0xffd0 && m <= 0xffd7) {
			*rst = r.pos - 2;
			return RET_SUCCESS;
		}
		if (m == 0xffd9) {
			return RET_SUCCESS;
		}
		const bool extra = tolerant && (m == 0xffe9 || m == 0xffea || m == 0xffef || m == 0xffdc || (m >= 0xfff0 && m <= 0xfffd));
		if ((m >= 0xffe0 && m <= 0xffe8) || (m >= 0xffeb && m <= 0xffee) || m == 0xffcc || m == 0xfffe || m == 0xffdd || extra) {
			if (!r.ok(2)) {
				return RET_FAILURE_FILE_IO;
			}
			const size_t len = ((size_t)p[r.pos] << 8) | p[r.pos + 1];
			if (m == 0xfffe && len < 2) {
				return RET_FAILURE_FILE_UNSUPPORTED;
			}
			if (m == 0xffdd) {
				if (!r.ok(4)) {
					return RET_FAILURE_FILE_IO;                    // read_length + read_word (decoder.c:616-619)
				}
				r.pos += 4;                                        // Ri is stored and never used (quirk Q1)
				continue;
			}
			r.pos += len;
			if (r.pos > n) {
				return RET_FAILURE_FILE_IO;
			}
			continue;
		}
		return RET_FAILURE_FILE_UNSUPPORTED;                   // tables, frames, scans, unknown markers: not on this path
	}
}

// read_ecs_byte (io.c:247-274): the entropy-coded data ends at the first FF xx with xx not in {00, D0..D7}
size_t jh_ecs_end(const uint8_t *p, size_t n, size_t pos)
{
	while (pos < n) {
		const uint8_t *q = (const uint8_t *)memchr(p + pos, 0xff, n - pos);
		if (!q) {
			return n;
		}
		pos = (size_t)(q - p);
		if (pos + 1 >= n) {
			return n;
		}
		const unsigned x = p[pos + 1];
		if (x != 0x00 && !(x >= 0xd0 && x <= 0xd7)) {
			return pos;
		}
		pos += 2;
	}
	return n;
}

// ---- header producer ---------------------------------------------------------------------------------

static void put16(std::vector<uint8_t> *o, unsigned v)
{
	o->push_back((uint8_t)(v >> 8));
	o->push_back((uint8_t)v);
}

void jh_produce_headers(const JhGeom &g, const uint16_t Q[4][64], const JhHuffSpec ht[2][4],
                        int restart_interval, std::vector<uint8_t> *o)
{
	o->clear();
	put16(o, 0xffd8);                                          // SOI
	const int ntab = g.Nf > 1 ? 2 : 1;
	for (int t = 0; t < ntab; ++t) {                           // DQT: always Pq = 0 (encoder.c:205-234)
		put16(o, 0xffdb);
		put16(o, 67);
		o->push_back((uint8_t)t);
		for (int i = 0; i < 64; ++i) {
			o->push_back((uint8_t)Q[t][jh_zigzag[i]]);
		}
	}
	put16(o, 0xffc0);                                          // SOF0 (encoder.c:236-274)
	put16(o, 8u + 3u * (unsigned)g.Nf);
	o->push_back((uint8_t)g.P);
	put16(o, (unsigned)g.Y);
	put16(o, (unsigned)g.X);
	o->push_back((uint8_t)g.Nf);
	for (int k = 0; k < g.ncomp; ++k) {
		o->push_back((uint8_t)g.id[k]);
		o->push_back((uint8_t)((g.H[k] << 4) | g.V[k]));
		o->push_back((uint8_t)g.Tq[k]);
	}
	for (int th = 0; th < ntab; ++th) {                        // DHT: DC then AC per id (encoder.c:620-630)
		for (int tc = 0; tc < 2; ++tc) {
			const JhHuffSpec &s = ht[tc][th];
			unsigned mt = 0;
			for (int i = 0; i < 16; ++i) {
				mt += s.L[i];
			}
			put16(o, 0xffc4);
			put16(o, 2 + 17 + mt);
			o->push_back((uint8_t)((tc << 4) | th));
			o->insert(o->end(), s.L, s.L + 16);
			o->insert(o->end(), s.V, s.V + mt);
		}
	}
	if (restart_interval > 0) {                                // extension (absent from the reference)
		put16(o, 0xffdd);
		put16(o, 4);
		put16(o, (unsigned)restart_interval);
	}
	put16(o, 0xffda);                                          // SOS (encoder.c:345-400)
	put16(o, 6u + 2u * (unsigned)g.Nf);
	o->push_back((uint8_t)g.Nf);
	for (int k = 0; k < g.ncomp; ++k) {
		o->push_back((uint8_t)g.id[k]);
		o->push_back((uint8_t)((g.Td[k] << 4) | g.Ta[k]));
	}
	o->push_back(0);
	o->push_back(63);
	o->push_back(0);
}
