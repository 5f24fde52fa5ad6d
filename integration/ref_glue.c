/*
 * integration/ref_glue.c -- the binding a maintainer of xbarin02/jpeg adds to run the reference's OWN
 * drivers on libjpeg_amd.so.
 *
 * decoder.c and encoder.c stay as they are: main(), parse_format() (decoder.c:472-659),
 * produce_codestream() (encoder.c:589-645), every marker parser and producer, struct context.  This file
 * defines, with the reference's exact signatures, the five loops those drivers call:
 *
 *     read_ecs(FILE *, struct context *, struct scan *)          decoder.c:364
 *     epilogue(struct context *, const char *)                   decoder.c:456
 *     prologue(struct context *, FILE *, struct params *)        encoder.c:176
 *     write_ecs_dry(struct context *, struct scan *)             encoder.c:525
 *     write_ecs(FILE *, struct context *, struct scan *)         encoder.c:560
 *
 * by describing `struct context` to the stage-level C-ABI (include/jpeg_amd.h) and calling it.  The
 * reference's definitions of the five are weakened in the object files (oracle/Makefile: objcopy
 * --weaken-symbol), so the linker takes these; everything else of the reference links unchanged.
 *
 * Built by our own recipe into oracle/_ref/decoder_hip and oracle/_ref/encoder_hip; tests/test_gpu_ref_glue.py
 * runs them beside oracle/_ref/decoder and oracle/_ref/encoder and compares the files they write.
 *
 * Compile with -DGLUE_DECODER or -DGLUE_ENCODER (struct scan and the loops differ per program), with the
 * reference's directory on the include path.  This file holds no code of the reference.
 *
 * -DGLUE_STAGES is the other way to adopt the library, stage by stage (SURVEY.md section 8b's table): the reference
 * keeps ALL of its drivers and loops -- read_ecs(), epilogue(), prologue(), write_ecs() included -- and only its
 * arithmetic stages are replaced, under their own names and signatures:
 *
 *     int dequantize(struct context *)      imgproc.h:8      int quantize(struct context *)       imgproc.h:10
 *     int inverse_dct(struct context *)     imgproc.h:14     int forward_dct(struct context *)    imgproc.h:16
 *     int frame_to_rgb(struct frame *)      frame.h:21       int frame_to_ycc(struct frame *)     frame.h:33
 *
 * (weakened in imgproc.o / frame.o by the recipe; conv_blocks_to_frame(), conv_frame_to_blocks() and the two
 * transform_*() only move samples and stay the reference's own).  oracle/_ref/decoder_stages / encoder_stages.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <inttypes.h>
#include "common.h"   /* the reference's: struct context, component, qtable, htable */
#include "coeffs.h"   /* struct int_block */
#include "frame.h"    /* struct frame, read_frame_header() */
#include "huffman.h"  /* conv_htable_to_hcode() */
#include "jpeg_amd.h"

#ifndef JPEG_HIP_DEVICE
#define JPEG_HIP_DEVICE 0
#endif

#if defined(GLUE_DECODER) || defined(GLUE_ENCODER)
/* struct scan is private to decoder.c:186-195 / encoder.c:318-327 (the same three members in both) */
struct scan {
	uint8_t Ns;
	uint8_t Cs[256];
	struct int_block *last_block[256];
};

/* struct params is private to encoder.c:68-76; jpeg_hip_params has its layout */
struct params {
	uint8_t H, V;
	int q;
	int optimize;
};

/* struct context -> the flat descriptions of jpeg_amd.h.  Components in ascending identifier order, as
 * compute_no_blocks_and_alloc_buffers() walks them (common.c:163); index_of[] maps identifiers to that order. */
static int describe_frame(struct context *c, jpeg_hip_frame_desc *fd, int32_t *coef[JPEG_HIP_MAX_COMP],
                          uint16_t Q[4][64], int index_of[256])
{
	int k = 0;

	memset(fd, 0, sizeof *fd);
	fd->X = c->X;
	fd->Y = c->Y;
	fd->P = c->P;
	fd->Nf = c->Nf;
	for (int id = 0; id < 256; ++id) {
		index_of[id] = -1;
		if (c->component[id].H == 0) {
			continue;
		}
		if (k == JPEG_HIP_MAX_COMP) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		fd->H[k] = c->component[id].H;
		fd->V[k] = c->component[id].V;
		fd->Tq[k] = c->component[id].Tq;
		coef[k] = (int32_t *)c->component[id].int_buffer;      /* struct int_block = int32_t c[64] */
		index_of[id] = k++;
	}
	for (; k < JPEG_HIP_MAX_COMP; ++k) {
		coef[k] = NULL;
	}
	for (int t = 0; t < 4; ++t) {
		memcpy(Q[t], c->qtable[t].Q, sizeof Q[t]);             /* raster order (common.h:55) */
	}
	return RET_SUCCESS;
}

static int describe_scan(struct context *c, const struct scan *s, const int index_of[256], jpeg_hip_scan_desc *sd)
{
	memset(sd, 0, sizeof *sd);
	if (s->Ns < 1 || s->Ns > JPEG_HIP_MAX_COMP) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	sd->Ns = s->Ns;
	sd->restart_interval = c->Ri;
	for (int j = 0; j < s->Ns; ++j) {
		if (index_of[s->Cs[j]] < 0) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		sd->Cs[j] = index_of[s->Cs[j]];
		sd->Td[j] = c->component[s->Cs[j]].Td;
		sd->Ta[j] = c->component[s->Cs[j]].Ta;
	}
	return RET_SUCCESS;
}

/* struct htable (common.h:85-91: V[16][255], a row per code length) <-> the DHT payload (BITS, HUFFVAL) */
static void flatten_htable(const struct htable *h, jpeg_hip_htable *o)
{
	int n = 0;

	memset(o, 0, sizeof *o);
	memcpy(o->L, h->L, 16);
	for (int i = 0; i < 16; ++i) {
		for (int j = 0; j < h->L[i] && n < 256; ++j) {
			o->V[n++] = h->V[i][j];
		}
	}
}

#ifdef GLUE_ENCODER
static void unflatten_htable(const jpeg_hip_htable *o, struct htable *h)
{
	int n = 0;

	memset(h, 0, sizeof *h);
	memcpy(h->L, o->L, 16);
	for (int i = 0; i < 16; ++i) {
		for (int j = 0; j < o->L[i] && j < 255 && n < 256; ++j) {
			h->V[i][j] = o->V[n++];
		}
	}
}

#endif

static void describe_tables(struct context *c, jpeg_hip_htable dc[4], jpeg_hip_htable ac[4])
{
	for (int t = 0; t < 4; ++t) {
		flatten_htable(&c->htable[0][t], &dc[t]);
		flatten_htable(&c->htable[1][t], &ac[t]);
	}
}

#endif /* GLUE_DECODER || GLUE_ENCODER */

#ifdef GLUE_DECODER

/* read_ecs(), decoder.c:364-388.  The reference decodes one restart interval per call and comes back for the
 * next from parse_format()'s RSTn cases (decoder.c:622-633); the device decodes the whole scan, restart
 * intervals included, so the stream is left on the marker that ends the SCAN (where read_ecs_byte() would
 * leave it after the last interval, io.c:265) and the RSTn cases are never reached. */
int read_ecs(FILE *stream, struct context *context, struct scan *scan)
{
	jpeg_hip_frame_desc fd;
	jpeg_hip_scan_desc sd;
	jpeg_hip_htable dc[4], ac[4];
	int32_t *coef[JPEG_HIP_MAX_COMP];
	uint16_t Q[4][64];
	int index_of[256];
	int err;

	if (scan->Ns == 0) {
		return RET_SUCCESS;                                    /* read_macroblock(): nothing to do (decoder.c:271-273) */
	}
	if (scan->Ns > 1 && context->m_x == 0) {
		return RET_FAILURE_FILE_UNSUPPORTED;                   /* SOS before SOF (decoder.c:306-309) */
	}
	err = describe_frame(context, &fd, coef, Q, index_of);
	RETURN_IF(err);
	err = describe_scan(context, scan, index_of, &sd);
	RETURN_IF(err);
	describe_tables(context, dc, ac);

	/* the entropy-coded data has to be in memory: everything from here to the end of the file */
	long at = ftell(stream);
	if (at < 0 || fseek(stream, 0, SEEK_END) != 0) {
		return RET_FAILURE_FILE_SEEK;
	}
	long end = ftell(stream);
	if (end < at || fseek(stream, at, SEEK_SET) != 0) {
		return RET_FAILURE_FILE_SEEK;
	}
	size_t n = (size_t)(end - at), used = 0;
	uint8_t *ecs = malloc(n ? n : 1);
	if (ecs == NULL) {
		return RET_FAILURE_MEMORY_ALLOCATION;
	}
	if (fread(ecs, 1, n, stream) != n) {
		free(ecs);
		return RET_FAILURE_FILE_IO;
	}

	/* only the scan's components are written */
	int32_t *scan_coef[JPEG_HIP_MAX_COMP] = { NULL, NULL, NULL, NULL };
	for (int j = 0; j < sd.Ns; ++j) {
		scan_coef[sd.Cs[j]] = coef[sd.Cs[j]];
	}
	err = jpeg_hip_read_scan(JPEG_HIP_DEVICE, &fd, &sd, dc, ac, ecs, n, scan_coef, &used);
	free(ecs);
	if (fseek(stream, at + (long)used, SEEK_SET) != 0) {
		return RET_FAILURE_FILE_SEEK;
	}
	RETURN_IF(err);

	printf("Processed: scan of %i component(s) on HIP device %i\n", sd.Ns, JPEG_HIP_DEVICE);

	return RET_SUCCESS;
}

/* epilogue(), decoder.c:456-470: dequantize, inverse_dct, conv_blocks_to_frame, then write_image() =
 * frame_create + frame_to_rgb + write_frame (decoder.c:433-454).  The device returns the PPM/PGM body;
 * the header is the reference's own write_frame_header() (frame.c:407-429). */
int write_frame_header(struct frame *frame, int components, FILE *stream);

int epilogue(struct context *context, const char *path)
{
	jpeg_hip_frame_desc fd;
	int32_t *coef[JPEG_HIP_MAX_COMP];
	uint16_t Q[4][64];
	int index_of[256];
	int err;

	err = describe_frame(context, &fd, coef, Q, index_of);
	RETURN_IF(err);

	int comps = fd.Nf == 1 ? 1 : 3;                            /* write_frame(): 3 and 4 components -> P6 (frame.c:554-561) */
	size_t body_size = (size_t)fd.X * fd.Y * comps * (fd.P > 8 ? 2 : 1);
	void *body = malloc(body_size ? body_size : 1);
	if (body == NULL) {
		return RET_FAILURE_MEMORY_ALLOCATION;
	}
	err = jpeg_hip_decode_pixels(JPEG_HIP_DEVICE, &fd, (const uint16_t (*)[64])Q, (const int32_t *const *)coef, body);
	if (err) {
		free(body);
		return err;
	}

	if (path == NULL) {
		path = comps == 1 ? "output.pgm" : "output.ppm";
	}
	FILE *o = fopen(path, "w");
	if (o == NULL) {
		free(body);
		return RET_FAILURE_FILE_OPEN;
	}
	struct frame frame;
	memset(&frame, 0, sizeof frame);
	frame.components = fd.Nf;
	frame.X = context->X;
	frame.Y = context->Y;
	frame.precision = context->P;
	err = write_frame_header(&frame, comps, o);
	if (!err && fwrite(body, 1, body_size, o) < body_size) {
		err = RET_FAILURE_FILE_IO;
	}
	fclose(o);
	free(body);

	return err;
}

#endif /* GLUE_DECODER */

#ifdef GLUE_ENCODER

/* prologue(), encoder.c:176-193: read_image() (encoder.c:90-173) + conv_frame_to_blocks + forward_dct +
 * quantize.  Header and set-up on the host, with the reference's own read_frame_header() (frame.c:448) and
 * compute_no_blocks_and_alloc_buffers() (common.c:156); the body goes to the device as it lies in the file. */
int prologue(struct context *context, FILE *i_stream, struct params *params)
{
	struct frame frame;
	int err;

	err = read_frame_header(&frame, i_stream);
	RETURN_IF(err);

	printf("read PPM/PGM header: Nf=%" PRIu8 " Y=%" PRIu16 " X=%" PRIu16 " P=%" PRIu8 "\n", frame.components, frame.Y, frame.X, frame.precision);

	context->Nf = frame.components;
	context->Y = frame.Y;
	context->X = frame.X;
	context->P = frame.precision;

	/* component set-up of read_image() (encoder.c:109-149): identifiers 1..3, luma H x V from -h / -v,
	 * chroma 1 x 1, quantisation and Huffman tables 0 / 1 / 1 */
	if (frame.components != 1 && frame.components != 3) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	for (int id = 1; id <= frame.components; ++id) {
		struct component *comp = &context->component[id];
		int luma = id == 1;
		comp->H = luma && frame.components == 3 ? params->H : 1;
		comp->V = luma && frame.components == 3 ? params->V : 1;
		comp->Tq = comp->Td = comp->Ta = luma ? 0 : 1;
	}
	context->max_H = context->component[1].H;
	context->max_V = context->component[1].V;

	jpeg_hip_set_qtable(context->qtable[0].Q, 0, params->q);   /* set_qtable() x 2 (encoder.c:151-152) */
	jpeg_hip_set_qtable(context->qtable[1].Q, 1, params->q);

	err = compute_no_blocks_and_alloc_buffers(context);
	RETURN_IF(err);

	size_t body_size = (size_t)frame.X * frame.Y * frame.components * (frame.precision > 8 ? 2 : 1);
	void *body = malloc(body_size ? body_size : 1);
	if (body == NULL) {
		return RET_FAILURE_MEMORY_ALLOCATION;
	}
	if (fread(body, 1, body_size, i_stream) < body_size) {
		free(body);
		return RET_FAILURE_FILE_IO;                            /* read_frame_body() (frame.c:296-299) */
	}

	jpeg_hip_frame_desc fd;
	int32_t *coef[JPEG_HIP_MAX_COMP];
	uint16_t Q[4][64];
	int index_of[256];
	err = describe_frame(context, &fd, coef, Q, index_of);
	if (!err) {
		err = jpeg_hip_encode_pixels(JPEG_HIP_DEVICE, &fd, (const uint16_t (*)[64])Q, body, coef);
	}
	free(body);

	return err;
}

/* write_ecs_dry(), encoder.c:525-558: statistics + adapt_huffman_table() + conv_htable_to_hcode() for the
 * tables in use; the result lands in context->htable[][] where produce_DHT() reads it (encoder.c:620-630). */
int write_ecs_dry(struct context *context, struct scan *scan)
{
	jpeg_hip_frame_desc fd;
	jpeg_hip_scan_desc sd;
	jpeg_hip_htable dc[4], ac[4];
	int32_t *coef[JPEG_HIP_MAX_COMP];
	uint16_t Q[4][64];
	int index_of[256];
	int err;

	err = describe_frame(context, &fd, coef, Q, index_of);
	RETURN_IF(err);
	err = describe_scan(context, scan, index_of, &sd);
	RETURN_IF(err);
	sd.restart_interval = 0;                                   /* the reference encoder writes no restart markers */
	describe_tables(context, dc, ac);

	err = jpeg_hip_adapt_tables(JPEG_HIP_DEVICE, &fd, &sd, dc, ac, (const int32_t *const *)coef);
	RETURN_IF(err);

	for (int j = 0; j < 2; ++j) {
		for (int i = 0; i < (context->Nf > 1 ? 2 : 1); ++i) {
			unflatten_htable(j == 0 ? &dc[i] : &ac[i], &context->htable[j][i]);
			err = conv_htable_to_hcode(&context->htable[j][i], &context->hcode[j][i]);
			RETURN_IF(err);
		}
	}
	context->mblocks = context->m_x * context->m_y;

	return RET_SUCCESS;
}

/* write_ecs(), encoder.c:560-587: the entropy-coded segment between the SOS header and EOI */
int write_ecs(FILE *stream, struct context *context, struct scan *scan)
{
	jpeg_hip_frame_desc fd;
	jpeg_hip_scan_desc sd;
	jpeg_hip_htable dc[4], ac[4];
	int32_t *coef[JPEG_HIP_MAX_COMP];
	uint16_t Q[4][64];
	int index_of[256];
	int err;

	err = describe_frame(context, &fd, coef, Q, index_of);
	RETURN_IF(err);
	err = describe_scan(context, scan, index_of, &sd);
	RETURN_IF(err);
	sd.restart_interval = 0;
	describe_tables(context, dc, ac);

	/* upper bound: 64 coefficients of at most 16 + 16 bits, every byte stuffed */
	size_t blocks = 0;
	for (int id = 0; id < 256; ++id) {
		blocks += context->component[id].b_x * context->component[id].b_y;
	}
	size_t cap = blocks * 64 * 4 * 2 + 4096, n = 0;
	uint8_t *out = malloc(cap);
	if (out == NULL) {
		return RET_FAILURE_MEMORY_ALLOCATION;
	}
	err = jpeg_hip_write_scan(JPEG_HIP_DEVICE, &fd, &sd, dc, ac, 0, (const int32_t *const *)coef, out, cap, &n);
	if (!err && fwrite(out, 1, n, stream) < n) {
		err = RET_FAILURE_FILE_IO;
	}
	free(out);
	RETURN_IF(err);

	context->mblocks = context->m_x * context->m_y;
	printf("Processed: %zu macroblocks\n", context->mblocks);

	return RET_SUCCESS;
}

#endif /* GLUE_ENCODER */

#ifdef GLUE_STAGES
/* ---- the arithmetic stages, one GPU call per component (or per frame) each ------------------------------ */

int dequantize(struct context *context)                        /* imgproc.c:32-56 */
{
	for (int i = 0; i < 256; ++i) {
		struct component *c = &context->component[i];
		if (c->int_buffer != NULL) {
			int err = jpeg_hip_dequantize_blocks(JPEG_HIP_DEVICE, c->b_x * c->b_y, (const int32_t *)c->int_buffer,
			                                     context->qtable[c->Tq].Q, (float *)c->flt_buffer);
			RETURN_IF(err);
		}
	}
	return RET_SUCCESS;
}

int quantize(struct context *context)                          /* imgproc.c:58-82 */
{
	for (int i = 0; i < 256; ++i) {
		struct component *c = &context->component[i];
		if (c->int_buffer != NULL) {
			int err = jpeg_hip_quantize_blocks(JPEG_HIP_DEVICE, c->b_x * c->b_y, (const float *)c->flt_buffer,
			                                   context->qtable[c->Tq].Q, (int32_t *)c->int_buffer);
			RETURN_IF(err);
		}
	}
	return RET_SUCCESS;
}

int inverse_dct(struct context *context)                       /* imgproc.c:172-200 */
{
	for (int i = 0; i < 256; ++i) {
		struct component *c = &context->component[i];
		if (c->int_buffer != NULL) {
			int err = jpeg_hip_inverse_dct_blocks(JPEG_HIP_DEVICE, c->b_x * c->b_y, (float *)c->flt_buffer, context->P);
			RETURN_IF(err);
		}
	}
	return RET_SUCCESS;
}

int forward_dct(struct context *context)                       /* imgproc.c:202-230 */
{
	for (int i = 0; i < 256; ++i) {
		struct component *c = &context->component[i];
		if (c->int_buffer != NULL) {
			int err = jpeg_hip_forward_dct_blocks(JPEG_HIP_DEVICE, c->b_x * c->b_y, (float *)c->flt_buffer, context->P);
			RETURN_IF(err);
		}
	}
	return RET_SUCCESS;
}

int frame_to_rgb(struct frame *frame)                          /* frame.c:188-244 */
{
	return jpeg_hip_frame_to_rgb(JPEG_HIP_DEVICE, frame->data, frame->components, frame->X, frame->Y,
	                             frame->size_x, frame->size_y, frame->precision);
}

int frame_to_ycc(struct frame *frame)                          /* frame.c:154-186 */
{
	return jpeg_hip_frame_to_ycc(JPEG_HIP_DEVICE, frame->data, frame->components, frame->X, frame->Y,
	                             frame->size_x, frame->size_y, frame->precision);
}
#endif /* GLUE_STAGES */
