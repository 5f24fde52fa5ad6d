/*
 * oracle/ref_tap_dec.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Tap harness linked with the UNMODIFIED reference decoder objects (compiled in
 * place from /root/reference by oracle/Makefile).  It drives the reference's own
 * entry point process_jpeg_stream() (decoder.c:661) on an in-memory JPEG and, through
 * ld --wrap, observes two stage boundaries of the unmodified flow:
 *   - dequantize()  (imgproc.c:32, called from epilogue decoder.c:460): on entry the
 *     per-component int_buffer holds the entropy-decoded, DC-unpredicted coefficients;
 *   - write_frame() (frame.c:548, called from write_image decoder.c:448): on entry
 *     frame->data holds the float samples right before roundf/clamp.
 * Nothing in here is part of the product; it only exists so that tests can pin
 * oracle/jpeg_oracle.c (our restatement) against the real reference.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include "common.h"
#include "coeffs.h"
#include "imgproc.h"
#include "frame.h"

int process_jpeg_stream(FILE *stream, const char *path);
int __real_dequantize(struct context *context);
int __real_write_frame(struct frame *frame, const char *path);
extern float lut[8][8];
void init_lut(void);

#define TAP_MAXC 8
static struct {
	int n;
	int id[TAP_MAXC], H[TAP_MAXC], V[TAP_MAXC], Tq[TAP_MAXC];
	size_t bx[TAP_MAXC], by[TAP_MAXC];
	int32_t *coef[TAP_MAXC];
	int P, X, Y, Nf;
	uint16_t Q[4][64];
	/* frame tap */
	float *fdata;
	size_t size_x, size_y;
	int fcomp;
} tap;

static void tap_reset(void)
{
	for (int k = 0; k < TAP_MAXC; ++k) {
		free(tap.coef[k]);
	}
	free(tap.fdata);
	memset(&tap, 0, sizeof tap);
}

int __wrap_dequantize(struct context *context)
{
	tap.P = context->P; tap.X = context->X; tap.Y = context->Y; tap.Nf = context->Nf;
	for (int t = 0; t < 4; ++t) {
		memcpy(tap.Q[t], context->qtable[t].Q, sizeof tap.Q[t]);
	}
	for (int i = 0; i < 256 && tap.n < TAP_MAXC; ++i) {
		struct component *c = &context->component[i];
		if (c->int_buffer != NULL) {
			int k = tap.n++;
			size_t blocks = c->b_x * c->b_y;
			tap.id[k] = i; tap.H[k] = c->H; tap.V[k] = c->V; tap.Tq[k] = c->Tq;
			tap.bx[k] = c->b_x; tap.by[k] = c->b_y;
			tap.coef[k] = malloc(blocks * 64 * sizeof(int32_t));
			memcpy(tap.coef[k], c->int_buffer, blocks * 64 * sizeof(int32_t));
		}
	}
	return __real_dequantize(context);
}

int __wrap_write_frame(struct frame *frame, const char *path)
{
	size_t n = (size_t)frame->components * frame->size_x * frame->size_y;
	tap.fdata = malloc(n * sizeof(float));
	memcpy(tap.fdata, frame->data, n * sizeof(float));
	tap.size_x = frame->size_x; tap.size_y = frame->size_y; tap.fcomp = frame->components;
	return __real_write_frame(frame, path);
}

/* silence the reference's progress printf()s while it runs */
static int quiet_begin(void)
{
	fflush(stdout);
	int saved = dup(1);
	int nul = open("/dev/null", O_WRONLY);
	dup2(nul, 1);
	close(nul);
	return saved;
}

static void quiet_end(int saved)
{
	fflush(stdout);
	dup2(saved, 1);
	close(saved);
}

/* Decode an in-memory JPEG with the reference; PPM/PGM goes to out_path. */
int ref_dec_run(const uint8_t *jpg, size_t n, const char *out_path, int quiet)
{
	tap_reset();
	FILE *f = fmemopen((void *)jpg, n, "r");
	if (f == NULL) {
		return -2;
	}
	int saved = quiet ? quiet_begin() : -1;
	int err = process_jpeg_stream(f, out_path);
	if (quiet) {
		quiet_end(saved);
	}
	fclose(f);
	return err;
}

int ref_tap_ncomp(void) { return tap.n; }

void ref_tap_frame_info(int *P, int *X, int *Y, int *Nf)
{
	*P = tap.P; *X = tap.X; *Y = tap.Y; *Nf = tap.Nf;
}

void ref_tap_comp_info(int k, int *id, int *H, int *V, int *Tq, size_t *bx, size_t *by)
{
	*id = tap.id[k]; *H = tap.H[k]; *V = tap.V[k]; *Tq = tap.Tq[k];
	*bx = tap.bx[k]; *by = tap.by[k];
}

const int32_t *ref_tap_coef(int k) { return tap.coef[k]; }

const uint16_t *ref_tap_qtable(int t) { return tap.Q[t]; }

const float *ref_tap_frame(size_t *size_x, size_t *size_y, int *comps)
{
	*size_x = tap.size_x; *size_y = tap.size_y; *comps = tap.fcomp;
	return tap.fdata;
}

/* the reference's DCT look-up table (imgproc.c:93-102), for pinning table T1 */
void ref_dump_lut(float out[64])
{
	init_lut();
	memcpy(out, lut, sizeof(float) * 64);
}

/* direct calls into the reference's block primitives (imgproc.c:130,151,10,21) for KATs */
void idct(struct flt_block *flt_block);
void fdct(struct flt_block *flt_block);
void quantize_block(struct int_block *int_block, struct flt_block *flt_block, struct qtable *qtable);

void ref_idct_block(float c[64]) { idct((struct flt_block *)c); }

void ref_fdct_block(float c[64]) { fdct((struct flt_block *)c); }

void ref_quantize_block(int32_t out[64], const float in[64], const uint16_t Q[64])
{
	struct qtable qt;
	struct flt_block fb;
	qt.Pq = 0;
	memcpy(qt.Q, Q, sizeof qt.Q);
	memcpy(fb.c, in, sizeof fb.c);
	quantize_block((struct int_block *)out, &fb, &qt);
}

void ref_dequantize_block(float out[64], const int32_t in[64], const uint16_t Q[64])
{
	struct qtable qt;
	struct int_block ib;
	qt.Pq = 0;
	memcpy(qt.Q, Q, sizeof qt.Q);
	memcpy(ib.c, in, sizeof ib.c);
	dequantize_block(&ib, (struct flt_block *)out, &qt);
}
