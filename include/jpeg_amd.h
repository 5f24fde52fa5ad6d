/*
 * include/jpeg_amd.h -- C-ABI of libjpeg_amd.so, the MI355X (gfx950) hot path of a
 * baseline-JPEG codec that keeps the semantics of xbarin02/jpeg.
 *
 * The reference has no plugin/FFI layer: its boundary is the internal C API on
 * `struct context` / `struct frame`.  The hot path replaced here is
 *
 *   decode:  read_ecs() loop            decoder.c:364-388  (+ read_macroblock :262-362,
 *                                        read_block coeffs.c:196-254, read_code huffman.c:255-271,
 *                                        next_bit/read_ecs_byte io.c:18-41,247-274)
 *            epilogue()                 decoder.c:456-470  (dequantize imgproc.c:32, inverse_dct :172,
 *                                        conv_blocks_to_frame :233, frame_create frame.c:134,
 *                                        frame_to_rgb :188, write_frame_body :352)
 *   encode:  prologue()                 encoder.c:176-193  (read_frame_body frame.c:277, frame_to_ycc :154,
 *                                        transform_frame_to_components :84, conv_frame_to_blocks
 *                                        imgproc.c:264, forward_dct :202, quantize :58)
 *            write_ecs_dry()/write_ecs() encoder.c:525-587 (+ write_block[_dry] coeffs.c:256-363,
 *                                        write_code huffman.c:274, put_bit/flush_bits io.c:43-87)
 *
 * Everything is plain C: pointers and sizes, no C++ or torch types.  Device pointers are
 * ordinary `void *` obtained from hipMalloc (or torch); `stream` is a hipStream_t passed as
 * `void *` (NULL = the default stream).  All functions return the reference's RET_* codes
 * (common.h:15-31).  The library fails loudly (RET_FAILURE_LOGIC_ERROR + message on stderr)
 * when no HIP device is usable: there is NO CPU fallback.
 */
#ifndef JPEG_AMD_H
#define JPEG_AMD_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: identical values to the reference (common.h:15-31) ------------------
 * A translation unit of the reference that includes its own common.h first (integration/ref_glue.c)
 * already has them. */
#ifndef JPEG_COMMON_H
enum {
	RET_SUCCESS                   = 0x0000,
	RET_FAILURE_FILE_IO           = 0x1000,
	RET_FAILURE_FILE_UNSUPPORTED  = 0x1001,
	RET_FAILURE_FILE_OPEN         = 0x1002,
	RET_FAILURE_FILE_SEEK         = 0x1003,
	RET_FAILURE_MEMORY_ALLOCATION = 0x2000,
	RET_FAILURE_LOGIC_ERROR       = 0x3000,
	RET_FAILURE_OVERFLOW_ERROR    = 0x3001,
	RET_FAILURE_NO_MORE_DATA      = 0x4000
};
#endif
/* device-side conditions for which the reference assert()s or returns -1:
 *   JPEG_HIP_ERR_DC_RANGE  |DC difference| > 2047               (assert, encoder.c:447)
 *   JPEG_HIP_ERR_NO_CODE   symbol absent from the Huffman table  (value_to_vlc == -1, huffman.c:252) */
#define JPEG_HIP_ERR_DC_RANGE 0x5001
#define JPEG_HIP_ERR_NO_CODE  (-1)

#define JPEG_HIP_MAX_COMP 4

/* Geometry of the frames of one batch (common.c:156-195, frame.c:16-35). */
typedef struct jpeg_hip_frame_info {
	int P, X, Y, Nf;                      /* precision, samples per line, lines, components */
	int id[JPEG_HIP_MAX_COMP];            /* component identifiers, ascending (the reference walks ids 0..255) */
	int H[JPEG_HIP_MAX_COMP], V[JPEG_HIP_MAX_COMP];
	int Tq[JPEG_HIP_MAX_COMP], Td[JPEG_HIP_MAX_COMP], Ta[JPEG_HIP_MAX_COMP];
	int max_H, max_V;
	int m_x, m_y;                         /* MCUs horizontally / vertically */
	int size_x, size_y;                   /* padded frame */
	int bx[JPEG_HIP_MAX_COMP], by[JPEG_HIP_MAX_COMP];   /* blocks per component */
	int Ri;                               /* restart interval announced by DRI (the reference ignores it) */
	int blocks_per_mcu;
	int out_components;                   /* 1 (PGM) or 3 (PPM): frame.c:548-567 */
	int sample_bytes;                     /* 1, or 2 (big-endian) when P > 8: frame.c:246-257 */
} jpeg_hip_frame_info;

/* ---- library ---------------------------------------------------------------------------- */

/* Number of usable HIP devices (0 = none: every entry point below then fails loudly). */
int jpeg_hip_device_count(void);
const char *jpeg_hip_version(void);
/* Bumped whenever a structure or a signature of this header changes incompatibly (2: jpeg_hip_params lost
 * restart_interval; 3 and 4: additions only -- 4 has jpeg_hip_decoder_set_lanes).  A binding compares jpeg_hip_abi_version() with the JPEG_HIP_ABI_VERSION it
 * was built against and refuses to go on when the library is older or has another layout of jpeg_hip_params. */
#define JPEG_HIP_ABI_VERSION 4
int jpeg_hip_abi_version(void);

/* Host only, no GPU needed: the headers of one codestream up to its first SOS, as parse_format() reads them
 * (decoder.c:472-601: SOI, APPn, DQT, SOF0/SOF1, DHT, DRI, SOS), and the geometry that follows from them
 * (common.c:156-195).  Returns what the reference returns on the same bytes: RET_FAILURE_FILE_UNSUPPORTED
 * for progressive / lossless / arithmetic frames and unknown markers, RET_FAILURE_FILE_IO for a file cut
 * inside the headers.  tolerant: see jpeg_hip_decoder_set_tolerant(). */
int jpeg_hip_parse_info(const uint8_t *jpg, size_t len, int tolerant, jpeg_hip_frame_info *info);

/* Self-test of the quantiser's arithmetic (no counterpart in the reference: quantize_block() simply writes
 * roundf(f / Q), imgproc.c:28).  The encode kernel splits the binary32 division into a per-table-entry part and
 * a per-coefficient part and rounds by add-and-truncate; this runs both against the compiler's `/` and roundf()
 * on the device for EVERY divisor 1 .. 65535, numerators on and next to `ties_per_divisor` rounding ties each
 * plus pseudo-random ones.  out[0] = cases tried, out[1] = quotients that differ (where at least one is >= 0.25 in
 * magnitude), out[2] = rounded results that differ.  Both must be zero. */
int jpeg_hip_selftest_divide(int device, int ties_per_divisor, unsigned seed, unsigned long long out[3]);

/* Self-test of the colour conversion for 8-bit samples (frame_to_ycc, frame.c:154-186, evaluates in binary64; the
 * encode kernel at precision 8 uses exact binary32 integers and one correctly rounded division per value,
 * jpeg_amd/csrc/ycc8.inc): both forms on the device for ALL 2^24 (R, G, B).  out[0] = triples tried,
 * out[1] / out[2] = triples on which the one-pixel / the packed two-pixel form differs from the binary64
 * expressions in any bit.  Both must be zero.  (The same enumeration against the compiled reference runs on the
 * host: tests/test_ycc8_cpu.py.) */
int jpeg_hip_selftest_ycc8(int device, unsigned long long out[3]);

/* ---- batch decoder ------------------------------------------------------------------------
 * One decoder object owns the device workspace for a batch of frames that share geometry
 * (Motion JPEG: every frame is an independent SOI..EOI codestream, decoder.c:603-612).
 *
 *   prepare  host side: parse_format() (decoder.c:472-659) restated on memory up to SOS, Annex C
 *            tables (huffman.c:150-178), upload of the codestreams.  Not part of the hot path.
 *   run      device side, asynchronous on `stream`: byte unstuffing + RSTn segmentation
 *            (io.c:247-274, decoder.c:622-633), Huffman decode (speculative, self-synchronising
 *            sub-sequences; per restart interval where markers exist), DC un-prediction
 *            (decoder.c:351-355), then the fused epilogue (dequantise, IDCT, level shift, up-sampling
 *            by replication, YCbCr/YCCK -> RGB, roundf, clamp) writing the PPM/PGM body.
 *   sync     waits for that run (an event, not the whole stream: work queued behind it is not waited
 *            for) and returns per-frame RET_* codes.
 */
typedef struct jpeg_hip_decoder jpeg_hip_decoder;

int  jpeg_hip_decoder_create(jpeg_hip_decoder **dec, int device);
void jpeg_hip_decoder_destroy(jpeg_hip_decoder *dec);
int  jpeg_hip_decoder_prepare(jpeg_hip_decoder *dec, int nframes,
                              const uint8_t *const *jpg, const size_t *jpg_len);
/* The same for codestreams that lie in page-locked host memory, close together -- a Motion JPEG file read
 * into one buffer from jpeg_hip_host_alloc() (or hipHostMalloc / hipHostRegister): the span of memory that
 * holds the frames goes to the device in one DMA from where it lies (no staging copy), queued on a stream of
 * the decoder's own; prepare returns without waiting and run() makes its stream wait.  The memory must stay
 * as it is until sync() has returned.  Frames scattered over memory (the span more than twice their bytes)
 * take the staged path of jpeg_hip_decoder_prepare().  The reference reads with fread() from a FILE*
 * (io.c:89-95); this is the memory-side replacement of that for callers that own their I/O buffers. */
int  jpeg_hip_decoder_prepare_pinned(jpeg_hip_decoder *dec, int nframes,
                                     const uint8_t *const *jpg, const size_t *jpg_len);
void *jpeg_hip_host_alloc(size_t bytes);              /* page-locked host memory, NULL on failure */
void jpeg_hip_host_free(void *p);
int  jpeg_hip_decoder_info(const jpeg_hip_decoder *dec, jpeg_hip_frame_info *info);
/* bytes one decoded frame occupies: X*Y*out_components*sample_bytes */
size_t jpeg_hip_decoder_frame_bytes(const jpeg_hip_decoder *dec);
int  jpeg_hip_decoder_run(jpeg_hip_decoder *dec, void *d_samples, size_t frame_stride, void *stream);
int  jpeg_hip_decoder_sync(jpeg_hip_decoder *dec, int *frame_status);
/* Stage tap for parity tests: coefficients of component k of one frame, in the reference's
 * int_buffer layout (by*bx blocks of 64 int32, raster order inside a block; coeffs.h:9-11). */
int  jpeg_hip_decoder_get_coefficients(jpeg_hip_decoder *dec, int frame, int k, int32_t *out);
/* Per-stage device time of the LAST run, measured with hipEvents on the run's stream.
 * names/ms arrays of capacity cap; returns the number of stages.  Enabled by
 * jpeg_hip_decoder_set_timing(dec, 1) (adds one event pair per kernel launch). */
void jpeg_hip_decoder_set_timing(jpeg_hip_decoder *dec, int on);
int  jpeg_hip_decoder_get_timing(jpeg_hip_decoder *dec, const char **names, float *ms, int cap);
/* Diagnostics of the speculative Huffman decoder (no counterpart in the reference, whose read_ecs()
 * is serial).  stats[0] = non-zero if the frame fell back to the serial walk (bit mask of causes), stats[1] = workgroups whose
 * entry had to be repaired, stats[2] = blocks decoded, stats[3] = restart segments.  Valid after sync().
 * set_debug: bit 0 forces the repair path on every other workgroup, bit 1 forces the serial walk
 * (tests: all three paths must give identical results). */
int  jpeg_hip_decoder_get_stats(const jpeg_hip_decoder *dec, int frame, uint32_t stats[4]);
/* Diagnostics (after sync): how many chains of the frame's first scan walked 0, 1, 2 ... further sub-sequences
 * before they fell in with a younger one (hist[cap - 1] collects the rest).  No counterpart in the reference. */
int  jpeg_hip_decoder_get_chain_steps(jpeg_hip_decoder *dec, int frame, uint32_t *hist, int cap);
void jpeg_hip_decoder_set_debug(jpeg_hip_decoder *dec, int flags);
/* parse_format() (decoder.c:472-659) ends in "unhandled marker" -> RET_FAILURE_FILE_UNSUPPORTED on APP9,
 * APP10, APP15, DNL and JPGn segments (decoder.c:654-656); prepare() does the same by default.  on != 0
 * (before prepare()): such segments are skipped by their length instead.  No counterpart in the reference. */
void jpeg_hip_decoder_set_tolerant(jpeg_hip_decoder *dec, int on);
/* on != 0: prepare() queues the upload of the codestreams on `stream` (the stream run() will be given: stream
 * order then takes the place of the event run() waits for) instead of a copy stream of the decoder's own.  For
 * pipelines that keep a decoder per stream -- jpeg_amd/mjpeg.py decode_stream: the upload of one batch runs beside
 * the kernels of the other decoder's batch, and so do the narrow ends of the two batches' kernels; the process
 * then has a busy stream per decoder and no copy streams (a process has four hardware queues).  prepare() waits
 * on that stream for the tables and cleared work areas it queues there (small; the stream should hold nothing but
 * this decoder's work) and returns with the DMA of the codestreams in flight.  The stream must outlive the
 * decoder's use of it.  No counterpart in the reference. */
void jpeg_hip_decoder_set_upload_stream(jpeg_hip_decoder *dec, void *stream, int on);
/* Lanes.  n = 1 ... 4 (1: as before).  run() lays the batch's frames out over n streams -- run()'s own and n - 1
 * of the decoder's, measured like jpeg_hip_streams_create() does to run beside `stream` (the stream run() will be
 * given) and beside each other -- so that the narrow kernels of one run of frames (scans, chain links, survivors:
 * a few waves per SIMD that wait on each other) execute beside the wide ones of the others.  Nothing else changes:
 * the lanes start behind what run()'s stream holds at the call, that stream waits for them, results and sync() are
 * the same.  4K 4:2:0 q90, 128 frames: 180 -> 204 Gpix/s with 3 lanes; 8K 12-bit gray q95: 92 -> 118 (DESIGN.md
 * section 5).  A process has four hardware queues: lanes beyond what is free share a queue and gain nothing, so
 * pipelines that already keep several decoders busy (jpeg_amd/mjpeg.py) stay at 1.  Returns the number of lanes
 * seen to run side by side (<= n), negative on error.  Call between batches (it waits for a run() in flight).
 * lane_streams: NULL, or n - 1 streams of the caller's to use instead of the decoder's own (a caller that lays
 * out the four queues itself with jpeg_hip_streams_create(): one for its encoder, the rest for this; they must
 * outlive their use here, and the return value is then n, unmeasured).  The environment variable JPEG_AMD_LANES
 * (1 ... 4) sets the lanes of decoders that were never told, at their first run().
 * No counterpart in the reference (decoder.c:603-612 decodes one frame at a time on one thread). */
int  jpeg_hip_decoder_set_lanes(jpeg_hip_decoder *dec, int n, void *stream, void *const *lane_streams);
/* compressed bytes actually consumed per frame (entropy-coded segment incl. RSTn), for roofline accounting */
int  jpeg_hip_decoder_ecs_bytes(const jpeg_hip_decoder *dec, size_t *ecs_len);

/* ---- streams for pipelines -----------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE (one thread, no device).  A process has four hardware queues, and HIP maps each
 * stream to one of them in an order that depends on how many streams the process has made before; two streams of
 * one queue do not overlap.  A pipeline that keeps several batches in flight (jpeg_amd/mjpeg.py: decode_stream,
 * encode_stream) asks here for n <= 4 streams that were MEASURED to run side by side (two small kernels that note
 * their start and end times), and -- use_avoid != 0 -- side by side with `avoid` too (the stream the caller's own
 * work runs on; NULL = the null stream).  Returns how many of the n streams are proven to be on queues of their
 * own (the rest, if the process has no free queue left, share), -1 on failure.  The streams are non-blocking
 * hipStream_t; destroy them with jpeg_hip_streams_destroy(). */
int  jpeg_hip_streams_create(int device, int n, void *avoid, int use_avoid, void **streams);
void jpeg_hip_streams_destroy(int device, int n, void **streams);

/* ---- Motion JPEG ---------------------------------------------------------------------------
 * "Supports Motion JPEG" in the reference means implicit Huffman tables (common.c:90-99); its
 * process_jpeg_stream() decodes ONE SOI..EOI codestream and reports what follows as garbage
 * (decoder.c:603-612).  This splitter finds every codestream of a concatenated stream so that the
 * frames can be handed to jpeg_hip_decoder_prepare() in batches, or sharded over GPUs (one process
 * per GPU, frame f -> rank by contiguous chunks; no collective on the data path).  Host only.
 * Returns the number of frames found (may exceed cap; only the first cap are stored), -1 on bad
 * arguments. */
int  jpeg_hip_mjpeg_split(const uint8_t *stream, size_t len, size_t *frame_off, size_t *frame_len, int cap);

/* ---- batch encoder ------------------------------------------------------------------------
 *   configure host side: read_image() set-up (encoder.c:104-152): component sampling, quality
 *             scaled Annex K.1 tables (encoder.c:38-65), default tables (mjpeg.h), workspace.
 *   run       device side: fused prologue (sample -> float, edge replication, RGB -> YCbCr, box-mean
 *             down-sampling, level shift, FDCT, quantise), then entropy coding: [histograms
 *             (write_ecs_dry) -> K.2 tables ->] per-block bit lengths -> prefix sums -> wavefront bit
 *             packing -> 0xFF stuffing (write_ecs).  Restart markers are an EXTENSION
 *             (jpeg_hip_encoder_set_restart_interval): the reference never emits DRI/RSTn.
 *   sync      waits, returns per-frame RET_* codes and total codestream lengths.
 */
typedef struct jpeg_hip_encoder jpeg_hip_encoder;

/* the reference's command line parameters: same members, same layout as `struct params` (encoder.c:68-76),
 * so that a `struct params *` of the reference's own main() can be passed straight through */
typedef struct jpeg_hip_params {
	uint8_t H, V;       /* luma sampling factors, 1 or 2 (-h, -v) */
	int q;              /* quality 1..100 (-q) */
	int optimize;       /* optimised Huffman tables (-o) */
} jpeg_hip_params;

int  jpeg_hip_encoder_create(jpeg_hip_encoder **enc, int device);
void jpeg_hip_encoder_destroy(jpeg_hip_encoder *enc);
int  jpeg_hip_encoder_configure(jpeg_hip_encoder *enc, int nframes, int X, int Y, int Nf, int P,
                                const jpeg_hip_params *params);
/* EXTENSION (the reference writes no DRI / RSTn): MCUs per restart interval, 0..65535,
 * 0 = none (default).  Read by the next configure(). */
void jpeg_hip_encoder_set_restart_interval(jpeg_hip_encoder *enc, int mcus);
/* set_qtable() with the Annex K.1 tables and quality_to_sf() (encoder.c:14-65): Q receives the luminance
 * (chroma == 0) or chrominance table scaled to quality q, raster order, as struct qtable holds it
 * (common.h:52-57).  Host only. */
void jpeg_hip_set_qtable(uint16_t Q[64], int chroma, int q);
int  jpeg_hip_encoder_info(const jpeg_hip_encoder *enc, jpeg_hip_frame_info *info);
/* d_samples: nframes PPM/PGM bodies (u8, or big-endian u16 when P > 8), frame_stride bytes apart */
int  jpeg_hip_encoder_run(jpeg_hip_encoder *enc, const void *d_samples, size_t frame_stride, void *stream);
int  jpeg_hip_encoder_sync(jpeg_hip_encoder *enc, int *frame_status, size_t *jpg_len);
/* Motion JPEG stream out (the encode direction).  The reference writes its one codestream
 * to a FILE* as it goes (produce_codestream() encoder.c:589-645 through write_ecs_byte() io.c:277-290); a batch
 * leaves the device as ONE run of bytes instead: run() as above, then the finished codestreams (SOI .. EOI) are
 * packed back to back in frame order into out[0 .. cap) by kernels on the same stream -- a Motion JPEG stream as
 * jpeg_hip_mjpeg_split() reads it.  `out` is device memory (the kernels pack straight into it) or page-locked host
 * memory (jpeg_hip_host_alloc / hipHostMalloc: the kernels pack into a staging buffer of cap bytes in HBM, and
 * sync_stream(), which knows the stream's length by then, brings exactly that many bytes over with one DMA on a
 * copy stream of the encoder's own); anything else is refused with RET_FAILURE_LOGIC_ERROR.  run_stream() waits for
 * nothing.  sync_stream() waits for that run (an event) and for its DMA, and returns the per-frame RET_* codes and
 * frame_off[nframes + 1]: frame f occupies out[frame_off[f] .. frame_off[f + 1]), frame_off[nframes] is the
 * stream's length.  A frame that failed takes no room.  A codestream that does not fit what is left of cap is
 * dropped with RET_FAILURE_OVERFLOW_ERROR, and so is every frame behind it (the stream has no holes). */
int  jpeg_hip_encoder_run_stream(jpeg_hip_encoder *enc, const void *d_samples, size_t frame_stride,
                                 void *out, size_t cap, void *stream);
int  jpeg_hip_encoder_sync_stream(jpeg_hip_encoder *enc, int *frame_status, size_t *frame_off);
/* complete codestream (SOI .. EOI) of one frame copied to host memory */
int  jpeg_hip_encoder_get_jpeg(jpeg_hip_encoder *enc, int frame, uint8_t *out, size_t cap, size_t *len);
/* device-resident codestream of one frame (valid until the next run) */
const void *jpeg_hip_encoder_device_jpeg(jpeg_hip_encoder *enc, int frame, size_t *len);
int  jpeg_hip_encoder_get_coefficients(jpeg_hip_encoder *enc, int frame, int k, int32_t *out);
/* Stage taps at the reference's FLOAT stage boundaries, for parity tests (the fused kernels never store them
 * otherwise).  d_floats = device memory the next run() also fills, NULL = off (default):
 *   decoder: frame->data behind frame_to_rgb() (frame.c:188-244), before roundf() / clamp (frame.c:375-380): per frame
 *            Y rows of X pixels of 1 (PGM) or 3 (PPM) floats -- the visible part of the reference's padded frame;
 *   encoder: component[].frame_buffer behind transform_frame_to_components() (frame.c:84-132), before the level shift of
 *            forward_dct() (imgproc.c:220-222): per frame the component planes one after the other, by*8 rows of bx*8
 *            floats each (jpeg_hip_encoder_info). */
void jpeg_hip_decoder_set_float_tap(jpeg_hip_decoder *dec, void *d_floats);
void jpeg_hip_encoder_set_float_tap(jpeg_hip_encoder *enc, void *d_floats);
void jpeg_hip_encoder_set_timing(jpeg_hip_encoder *enc, int on);
int  jpeg_hip_encoder_get_timing(jpeg_hip_encoder *enc, const char **names, float *ms, int cap);

/* ---- one frame, no device allocation ---------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  jpeg_hip_encoder_run() for ONE frame:
 * same kernels, same bytes, but every buffer beyond the samples and the codestream (coefficients, bit lengths,
 * prefix sums, the unstuffed stream, histograms, tables, state) lives in one fixed device array inside the
 * library (jpeg_hip_encode_noalloc_limit() bytes, 32 MiB, held per GPU until the process ends).  The call
 * allocates nothing (no hipMalloc / hipFree, no host-to-device copy: the tables travel as kernel arguments),
 * is asynchronous on `stream`, and refuses frames whose workspace does not fit with RET_FAILURE_OVERFLOW_ERROR
 * and a message (1920x1088 4:4:4 8-bit fits, 4K does not: use the encoder object).  No restart markers.
 *   d_samples  the PPM/PGM body (X*Y*Nf samples, u8, or big-endian u16 when P > 8), device memory
 *   d_jpeg     receives SOI .. EOI; cap bytes, at least 8192
 *   d_result   two 32-bit words in device memory: [0] the frame's RET / JPEG_HIP_ERR code as
 *              jpeg_hip_encoder_sync() reports it (RET_FAILURE_OVERFLOW_ERROR: cap or the fixed array was
 *              too small for this image's entropy-coded data), [1] the codestream length (0 on error)
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads) race -- serialise them. */
size_t jpeg_hip_encode_noalloc_limit(void);
/* Set-up, once per device, OUTSIDE the caller's no-allocation region: the runtime loads the library's code object on
 * the device here and the fixed array comes into being with it, and its address is looked up.  MANDATORY before a
 * call under HIP graph capture (that look-up is not a capturable operation; tests/test_zz_no_alloc.py captures and
 * replays the call); otherwise optional -- the first jpeg_hip_encode_noalloc() on a device does it itself. */
int jpeg_hip_encode_noalloc_init(int device);
int jpeg_hip_encode_noalloc(int device, const jpeg_hip_params *params, int X, int Y, int Nf, int P,
                            const void *d_samples, void *d_jpeg, size_t cap, uint32_t *d_result, void *stream);

/* ---- stage level: the four hot loops, one call each, on what `struct context` holds ------------
 * For a maintainer who keeps the reference's own marker parser and producer (parse_format()
 * decoder.c:472-659, produce_codestream() encoder.c:589-645) and swaps only the loops (SURVEY.md
 * section 8b).  One frame per call, HOST pointers in and out, synchronous; the arrays are the members of
 * `struct context` flattened (common.h:52-113): sampling factors and table selectors per component in
 * frame order, quantisation tables in raster order as qtable[].Q, Huffman tables as the DHT payload
 * (BITS, HUFFVAL), coefficients as component[].int_buffer (b_y*b_x blocks of 64 int32, raster order
 * inside a block).  These calls pay an upload and a download each: the batch objects above are the
 * fast path, these are the drop-in and the stage-parity path. */
typedef struct jpeg_hip_frame_desc {
	int X, Y, P, Nf;                          /* context->X, Y, precision, Nf */
	int H[JPEG_HIP_MAX_COMP], V[JPEG_HIP_MAX_COMP], Tq[JPEG_HIP_MAX_COMP];   /* component[].H, V, Tq */
} jpeg_hip_frame_desc;

typedef struct jpeg_hip_htable {              /* struct htable, common.h:85-91 */
	uint8_t L[16];                            /* codes of each length 1..16 */
	uint8_t V[256];                           /* symbols in code order */
} jpeg_hip_htable;

typedef struct jpeg_hip_scan_desc {           /* struct scan, decoder.c:186-195, encoder.c:318-327 */
	int Ns;                                   /* components in the scan */
	int Cs[JPEG_HIP_MAX_COMP];                /* their indices in the frame (0-based, frame order), scan header order */
	int Td[JPEG_HIP_MAX_COMP], Ta[JPEG_HIP_MAX_COMP];   /* table selectors of the j-th scan component
	                                             (component[Cs].Td / Ta, parse_scan_header() decoder.c:197-259) */
	int restart_interval;                     /* read: context->Ri (sizes the workspace only; as in the reference
	                                             the markers decide).  write: EXTENSION, MCUs per interval, 0 = none */
} jpeg_hip_scan_desc;

/* read_ecs(stream, context, scan), decoder.c:364-388 with its callees: decodes the entropy-coded data of
 * ONE scan.  ecs[0 .. n) starts behind the SOS header and must reach at least the marker that ends the
 * scan (else RET_FAILURE_FILE_IO, as read_marker() at end of file).  dc[]/ac[] = context->htable[0/1][0..3]
 * as they stand when the scan starts.  coef[k] (components of the scan only) receive the coefficients with
 * DC prediction undone (decoder.c:351-355); blocks the data does not reach are zero (common.c:129).
 * *consumed = offset of the marker that ended the scan (where read_ecs_byte() leaves the stream, io.c:265). */
int jpeg_hip_read_scan(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                       const jpeg_hip_htable dc[4], const jpeg_hip_htable ac[4],
                       const uint8_t *ecs, size_t n, int32_t *const coef[JPEG_HIP_MAX_COMP], size_t *consumed);

/* epilogue(context, path) without the file, decoder.c:456-470: dequantize, inverse_dct,
 * conv_blocks_to_frame, frame_create, frame_to_rgb, write_frame_body.  Q = context->qtable[0..3].Q.
 * samples receives the PPM/PGM body: Y*X*(1 or 3) samples, u8 or big-endian u16 (P > 8). */
int jpeg_hip_decode_pixels(int device, const jpeg_hip_frame_desc *frame, const uint16_t Q[4][64],
                           const int32_t *const coef[JPEG_HIP_MAX_COMP], void *samples);

/* prologue(context, stream, params) behind read_frame_header(), encoder.c:176-193: read_frame_body,
 * frame_to_ycc, transform_frame_to_components, conv_frame_to_blocks, forward_dct, quantize.  The frame
 * must be one the reference encoder sets up (encoder.c:109-146: one component, or three with chroma 1x1,
 * luma H, V in 1..2, Tq 0 / 1 / 1); otherwise RET_FAILURE_FILE_UNSUPPORTED. */
int jpeg_hip_encode_pixels(int device, const jpeg_hip_frame_desc *frame, const uint16_t Q[4][64],
                           const void *samples, int32_t *const coef[JPEG_HIP_MAX_COMP]);

/* write_ecs_dry() + adapt_huffman_table() + write_ecs(), encoder.c:525-587, huffman.c:508-537: the
 * entropy-coded data of the frame's one interleaved scan (all components, table selectors 0 / 1 / 1 as the
 * reference encoder uses them), byte-stuffed and padded, WITHOUT the EOI marker.  optimize == 0: codes from
 * dc[0..1] / ac[0..1]; optimize != 0: the tables are derived from the coefficients (Annex K.2) and
 * RETURNED in dc[] / ac[] for the caller's DHT segments.  Returns JPEG_HIP_ERR_DC_RANGE / _NO_CODE where the
 * reference asserts / fails, RET_FAILURE_OVERFLOW_ERROR when cap is too small (*n = bytes needed). */
int jpeg_hip_write_scan(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                        jpeg_hip_htable dc[4], jpeg_hip_htable ac[4], int optimize,
                        const int32_t *const coef[JPEG_HIP_MAX_COMP], uint8_t *out, size_t cap, size_t *n);

/* write_ecs_dry(context, scan), encoder.c:525-558: the statistics of write_block_dry() (coeffs.c:312-363) over
 * the frame's one interleaved scan and adapt_huffman_table() (huffman.c:508-537) for the tables in use (DC and
 * AC of id 0, and of id 1 when the frame has three components), RETURNED in dc[] / ac[]: what the reference
 * leaves in context->htable[][] for produce_DHT().  Follow with jpeg_hip_write_scan(optimize = 0). */
int jpeg_hip_adapt_tables(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                          jpeg_hip_htable dc[4], jpeg_hip_htable ac[4],
                          const int32_t *const coef[JPEG_HIP_MAX_COMP]);

/* ---- the stages one by one ---------------------------------------------------------------------
 * SURVEY.md section 8b's table, row by row: the reference's per-stage functions on `struct context` / `struct frame`,
 * on the buffers they work on (host pointers, one call = upload + one kernel + download; the fused kernels of the
 * batch objects are the fast path, these are the drop-in and the parity path for the FLOAT stage boundaries).
 * integration/ref_glue.c (-DGLUE_STAGES) defines dequantize(), inverse_dct(), forward_dct(), quantize(),
 * frame_to_rgb() and frame_to_ycc() with the reference's signatures on top of them; conv_blocks_to_frame() /
 * conv_frame_to_blocks() (imgproc.h:18-20) and the two transform_*() (frame.c:38-132) only move and copy samples and
 * stay the reference's own there.
 *   dequantize_blocks   dequantize() / dequantize_block()  imgproc.c:10-19,32-56: out = (float)(coef * Q), Q = qtable[Tq].Q
 *   quantize_blocks     quantize()   / quantize_block()    imgproc.c:21-30,58-82: out = (int32)roundf(in / Q)
 *   inverse_dct_blocks  inverse_dct()                      imgproc.c:172-200:     idct, then + 2^(P-1), in place
 *   forward_dct_blocks  forward_dct()                      imgproc.c:202-230:     - 2^(P-1), then fdct, in place
 *     (arrays of nblocks blocks of 64, raster order inside a block: component[].int_buffer / flt_buffer, coeffs.h:9-16)
 *   frame_to_rgb        frame.c:188-244 (3 or 4 components; 1: nothing to do)   on frame->data: size_y rows of size_x
 *   frame_to_ycc        frame.c:154-186 (3 components; 1: nothing to do)        pixels of `components` floats, only
 *                                                                               x < X, y < Y are converted */
int jpeg_hip_dequantize_blocks(int device, size_t nblocks, const int32_t *coef, const uint16_t Q[64], float *out);
int jpeg_hip_quantize_blocks(int device, size_t nblocks, const float *in, const uint16_t Q[64], int32_t *out);
int jpeg_hip_inverse_dct_blocks(int device, size_t nblocks, float *blocks, int P);
int jpeg_hip_forward_dct_blocks(int device, size_t nblocks, float *blocks, int P);
int jpeg_hip_frame_to_rgb(int device, float *data, int components, int X, int Y, size_t size_x, size_t size_y, int P);
int jpeg_hip_frame_to_ycc(int device, float *data, int components, int X, int Y, size_t size_x, size_t size_y, int P);

/* ---- single-image drop-ins for the reference's top-level entry points ------------------------
 * Same argument meaning and error behaviour as the functions they replace; the pixel and entropy
 * work runs on the GPU, marker parsing/production and PPM/PGM I/O on the host. */
int jpeg_hip_process_jpeg_stream(FILE *stream, const char *path);                 /* decoder.c:661 */
int jpeg_hip_process_jpeg_file(const char *i_path, const char *o_path);           /* decoder.c:687 */
int jpeg_hip_process_stream(FILE *i_stream, FILE *o_stream, const jpeg_hip_params *params); /* encoder.c:647 */

#ifdef __cplusplus
}
#endif

#endif /* JPEG_AMD_H */
