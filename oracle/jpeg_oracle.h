/*
 * oracle/jpeg_oracle.h -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * CPU restatement, in plain C99, of the algorithms of xbarin02/jpeg that lie on the
 * hot path named by BASELINE.json's north_star.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file.  It is pinned against the compiled reference
 * (oracle/_ref, built from /root/reference by oracle/Makefile) by tests/test_oracle_*.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

/* same numeric values as the reference's RET_* codes (common.h:15-31) */
enum {
	ORC_OK              = 0x0000,
	ORC_ERR_IO          = 0x1000,
	ORC_ERR_UNSUPPORTED = 0x1001,
	ORC_ERR_MEMORY      = 0x2000,
	ORC_ERR_LOGIC       = 0x3000,
	ORC_ERR_NO_MORE_DATA = 0x4000,
	/* the reference assert()s here (encoder.c:447); the oracle reports instead of aborting */
	ORC_ERR_DC_RANGE    = 0x5001,
	/* value_to_vlc() == -1 in the reference (huffman.c:252): symbol missing from the table */
	ORC_ERR_NO_CODE     = -1
};

#define ORC_MAXC 4

typedef struct {
	int id, H, V, Tq, Td, Ta;
	size_t bx, by;      /* blocks horizontally / vertically (common.c:181-182) */
	int32_t *coef;      /* bx*by blocks of 64, raster order inside a block (coeffs.h:9-11) */
} orc_comp;

typedef struct {
	int P, X, Y, Nf;
	int max_H, max_V;
	size_t m_x, m_y;        /* MCUs (common.c:171-172) */
	size_t size_x, size_y;  /* padded frame (frame.c:21-22) */
	int ncomp;
	orc_comp comp[ORC_MAXC];   /* ascending component id, as the reference iterates */
	uint16_t Q[4][64];         /* raster order (common.h:52-57) */
	uint8_t htL[2][4][16];     /* DHT BITS  [Tc][Th] */
	uint8_t htV[2][4][256];    /* DHT HUFFVAL, flattened */
	size_t mblocks;            /* MCUs processed in the last scan */
	int Ri;
	float *frame;              /* interleaved float samples before roundf/clamp (size_y*size_x*Nf) */
	float *plane[ORC_MAXC];    /* encode: component planes behind transform_frame_to_components (frame.c:84-132),
	                              by*8 rows of bx*8 floats, before the level shift */
} orc_image;

/* ---- whole-codec entry points ------------------------------------------------ */

/* decoder.c:661 process_jpeg_stream restated on memory.  *samples is the PPM/PGM body
 * (u8, or big-endian u16 when P > 8) of X*Y*(Nf==1?1:3) samples. */
int orc_decode(const uint8_t *jpg, size_t n, orc_image **img, uint8_t **samples, size_t *samples_len);

/* encoder.c:647 process_stream restated on memory.  body = PPM/PGM body bytes.
 * restart_interval (in MCUs, 0 = none) is an EXTENSION absent from the reference
 * (its encoder never emits DRI/RSTn). */
int orc_encode(const uint8_t *body, int X, int Y, int Nf, int P, int H, int V, int q,
               int optimize, int restart_interval,
               orc_image **img, uint8_t **jpg, size_t *jpg_len);

/* EXTENSION, test input generator: the same image in several scans (what the reference decoder reads scan
 * by scan, decoder.c:593-601).  split 0: one scan per component (Ns == 1 walk, decoder.c:274-302;
 * restart_interval in blocks), 1: {first} + {others}, 2: {all but last} + {last} (MCU walk restricted to the
 * scan's components, interval in MCUs); redefine != 0 (split 0) re-sends the Huffman tables under ids 0/0 in
 * front of every scan. */
int orc_encode_scans(const uint8_t *body, int X, int Y, int Nf, int P, int H, int V, int q,
                     int optimize, int restart_interval, int redefine, int split,
                     uint8_t **jpg, size_t *jpg_len);

void orc_image_free(orc_image *img);
void orc_free(void *p);

/* ---- accessors for ctypes ---------------------------------------------------- */
void orc_img_info(const orc_image *img, int *P, int *X, int *Y, int *Nf, int *ncomp,
                  size_t *m_x, size_t *m_y, size_t *size_x, size_t *size_y, size_t *mblocks);
void orc_img_comp(const orc_image *img, int k, int *id, int *H, int *V, int *Tq, int *Td, int *Ta,
                  size_t *bx, size_t *by);
const int32_t *orc_img_coef(const orc_image *img, int k);
const uint16_t *orc_img_qtable(const orc_image *img, int t);
const float *orc_img_frame(const orc_image *img);
const float *orc_img_plane(const orc_image *img, int k);
void orc_img_htable(const orc_image *img, int Tc, int Th, uint8_t L[16], uint8_t V[256]);

/* ---- block / table primitives (KATs) ----------------------------------------- */
void orc_lut(float out[64]);                              /* imgproc.c:93-102, literal T1 */
void orc_idct_block(float c[64]);                         /* imgproc.c:130-149 */
void orc_fdct_block(float c[64]);                         /* imgproc.c:151-170 */
void orc_quantize_block(int32_t out[64], const float in[64], const uint16_t Q[64]);   /* imgproc.c:21-30 */
void orc_dequantize_block(float out[64], const int32_t in[64], const uint16_t Q[64]); /* imgproc.c:10-19 */
int  orc_encode_cat(int32_t c);                           /* coeffs.c:56-74 */
int  orc_encode_extra(int32_t c, int cat);                /* coeffs.c:76-83 */
int32_t orc_decode_coeff(int cat, int extra);             /* coeffs.c:33-48 */
void orc_set_qtable(uint16_t Q[64], int chroma, int q);   /* encoder.c:38-65 */
void orc_frame_to_ycc(float *px, size_t n, int P);        /* frame.c:162-176: n (R, G, B) triples in place */
/* Annex C code generation (huffman.c:45-178): sizes/codes per symbol (EHUFSI/EHUFCO) */
int  orc_make_codes(const uint8_t L[16], const uint8_t V[256], uint8_t ehufsi[256], uint16_t ehufco[256]);
/* Annex K.2 optimiser (huffman.c:330-537): freq[257] (freq[256] must be 1) -> L, V */
void orc_adapt_table(const size_t freq_in[257], uint8_t L[16], uint8_t V[256]);
/* the four default (MJPEG / Annex K.3) tables of mjpeg.h:5-82 */
void orc_default_table(int Tc, int Th, uint8_t L[16], uint8_t V[256]);

#endif
