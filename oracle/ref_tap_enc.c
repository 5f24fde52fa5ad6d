/*
 * oracle/ref_tap_enc.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Tap harness linked with the UNMODIFIED reference encoder objects (compiled in
 * place from /root/reference by oracle/Makefile).  It drives the reference's own
 * entry point process_stream() (encoder.c:647) on an in-memory PPM/PGM and, through
 * ld --wrap=quantize, copies each component's int_buffer right after the reference's
 * quantize() (imgproc.c:58, called from prologue encoder.c:189) returned, i.e. the
 * quantised coefficients that the entropy coder is about to see.
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

/* encoder.c:68-76 (the struct is private to encoder.c; same layout restated for the call) */
struct params {
	uint8_t H, V;
	int q;
	int optimize;
};

int process_stream(FILE *i_stream, FILE *o_stream, struct params *params);
int __real_quantize(struct context *context);

#define TAP_MAXC 8
static struct {
	int n;
	int id[TAP_MAXC], H[TAP_MAXC], V[TAP_MAXC], Tq[TAP_MAXC];
	size_t bx[TAP_MAXC], by[TAP_MAXC];
	int32_t *coef[TAP_MAXC];
	float *plane[TAP_MAXC];       /* component[].frame_buffer: the planes behind transform_frame_to_components (frame.c:84-132) */
	uint16_t Q[4][64];
} tap;

static void tap_reset(void)
{
	for (int k = 0; k < TAP_MAXC; ++k) {
		free(tap.coef[k]);
		free(tap.plane[k]);
	}
	memset(&tap, 0, sizeof tap);
}

int __wrap_quantize(struct context *context)
{
	int err = __real_quantize(context);

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
			/* conv_frame_to_blocks() (imgproc.c:264-293) only read it: still what the down-sampling left */
			tap.plane[k] = malloc(blocks * 64 * sizeof(float));
			memcpy(tap.plane[k], c->frame_buffer, blocks * 64 * sizeof(float));
		}
	}
	return err;
}

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

/*
 * Encode an in-memory PPM/PGM (header + body) with the reference.
 * On success *out / *out_len hold a malloc()ed JPEG; the caller frees it with ref_free().
 */
int ref_enc_run(const uint8_t *pnm, size_t n, int H, int V, int q, int optimize,
                uint8_t **out, size_t *out_len, int quiet)
{
	struct params params;
	char *obuf = NULL;
	size_t osize = 0;

	tap_reset();
	params.H = (uint8_t)H; params.V = (uint8_t)V; params.q = q; params.optimize = optimize;

	FILE *fi = fmemopen((void *)pnm, n, "r");
	FILE *fo = open_memstream(&obuf, &osize);
	if (fi == NULL || fo == NULL) {
		return -2;
	}
	int saved = quiet ? quiet_begin() : -1;
	int err = process_stream(fi, fo, &params);
	if (quiet) {
		quiet_end(saved);
	}
	fclose(fi);
	fclose(fo);
	*out = (uint8_t *)obuf;
	*out_len = osize;
	return err;
}

void ref_free(void *p) { free(p); }

int ref_tap_ncomp(void) { return tap.n; }

void ref_tap_comp_info(int k, int *id, int *H, int *V, int *Tq, size_t *bx, size_t *by)
{
	*id = tap.id[k]; *H = tap.H[k]; *V = tap.V[k]; *Tq = tap.Tq[k];
	*bx = tap.bx[k]; *by = tap.by[k];
}

const int32_t *ref_tap_coef(int k) { return tap.coef[k]; }

const float *ref_tap_plane(int k) { return tap.plane[k]; }

const uint16_t *ref_tap_qtable(int t) { return tap.Q[t]; }

/* The reference's own frame_to_ycc() (frame.c:154-186) on a caller-made interleaved float frame of
 * Y rows of X (R, G, B) triples (no padding: size_x = X), in place. */
int ref_frame_to_ycc(float *data, int X, int Y, int precision)
{
	struct frame frame;

	memset(&frame, 0, sizeof frame);
	frame.components = 3;
	frame.X = (uint16_t)X;
	frame.Y = (uint16_t)Y;
	frame.size_x = (size_t)X;
	frame.size_y = (size_t)Y;
	frame.precision = (uint8_t)precision;
	frame.data = data;
	return frame_to_ycc(&frame);
}
