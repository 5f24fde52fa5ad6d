// jpeg_amd/csrc/kernels.hpp -- launchers of the HIP kernels (internal).
#pragma once

#include <hip/hip_runtime.h>
#include "common.hpp"

// optional per-kernel timing hook (hipEvents recorded on the launch stream by api.hip)
struct JhTimerHook {
	void (*begin)(void *ctx, const char *name, hipStream_t s);
	void (*end)(void *ctx, hipStream_t s);
	void *ctx;
};
#define JH_T_BEGIN(hook, name, s) do { if ((hook) && (hook)->ctx) (hook)->begin((hook)->ctx, name, s); } while (0)
#define JH_T_END(hook, s) do { if ((hook) && (hook)->ctx) (hook)->end((hook)->ctx, s); } while (0)

// ---- per-frame descriptors shared by the entropy kernels ---------------------------------------

struct JhDecFrame {               // written by the host in prepare()
	uint32_t raw_begin;           // first ECS byte (absolute offset in the raw codestream buffer)
	uint32_t raw_end;             // end of the frame's file in the raw buffer
	uint32_t cs_off;              // unstuffed stream of this frame (bytes, 16-byte aligned)
	uint32_t sub_off;             // first sub-sequence record of this frame
	uint32_t sub_cap;             // records reserved
	uint32_t seg_off;             // first entry of this frame in the restart-segment table
	uint32_t seg_cap;             // entries reserved (incl. sentinel)
	uint32_t chunk_off;           // first entry in the per-chunk count arrays
	uint32_t nchunk;
	uint32_t lut_off;             // first JhHuffLut of this frame
	uint32_t n_lut;
	uint32_t wg_off;              // first entry of this frame in the workgroup exit table
	uint8_t  blk_dc[JH_MAX_MCU_BLOCKS];   // lut slot per block of the MCU
	uint8_t  blk_ac[JH_MAX_MCU_BLOCKS];
};

enum {                            // JhDecState.status bits
	JH_ST_SEG_OVERFLOW = 1,       // more restart segments than reserved
	JH_ST_SERIAL = 2,             // the parallel synchronisation gave up on this frame: serial walk used
	JH_ST_COEF_RANGE = 4,         // P > 12: a DC coefficient left the 16 bits coefficients are stored in
};

struct JhDecState {               // written by the device
	uint32_t term_pos;            // absolute position of the marker that ends the ECS (or raw_end)
	uint32_t cs_len;              // bytes of unstuffed entropy-coded data
	uint32_t nseg;                // restart segments
	uint32_t status;
	uint32_t nblocks;             // blocks decoded (for diagnostics)
	uint32_t nrepaired;           // chains workgroups redone by k_huff_walk (for diagnostics)
	uint32_t serial_cause;        // why the serial walk was asked for: 2 entry states still
	                              // inconsistent after the walk rounds, 4 a walk had to decode too much by itself, 16 test hook
	uint32_t term_marker;         // second byte of the marker at term_pos (0xd9 = EOI)
	uint32_t ndrop;               // restart segments that ended inside an MCU (k_huff_fixup has work)
	uint32_t nsurv;               // chains that walk on after step 1 (k_huff_links -> k_huff_survivors)
	uint32_t ntail;               // the LAST segment ended inside an MCU: blocks below this index hold what the
	                              // reference leaves of it (decoder.c:347-355) and are not cleared
};

// A scan that codes ONE component of a multi-component frame (non-interleaved order, decoder.c:274-302)
// walks that component's blocks as a raster over its padded grid; a scan of SOME components
// (decoder.c:303-358) walks the frame's MCUs but only these components' blocks of each.  The coefficient
// buffer keeps the interleaved order the pixel kernel reads, so the write pass translates block numbers.
struct JhRemap {
	int mode;                     // 0: the scan order IS the buffer order; 1: raster of one component;
	                              // 2: MCUs of a subset of the components
	int bxc;                      // blocks per row of the component (b_x)
	int H, V;                     // its sampling factors
	int m_x;                      // MCUs per row of the frame
	int B;                        // blocks per MCU of the frame (buffer order)
	int kfirst;                   // first block of the component inside an MCU
	int nblk_full;                // blocks per frame in the buffer
	int nk;                       // mode 2: blocks per MCU of the scan
	uint8_t kmap[JH_MAX_MCU_BLOCKS];  // mode 2: block of the scan's MCU -> block of the frame's MCU
};

struct JhDecodeWork {             // device pointers of one decoder batch
	const uint8_t *raw;           // all codestreams
	uint8_t *cs;                  // unstuffed streams
	uint32_t *seg_start;          // byte offset of every restart segment inside cs (per frame slices)
	uint32_t *chunk_kept, *chunk_rst;
	JhSubInfo *rec;               // [sub-sequence t][step d <= JH_CHAIN_MAX]: chain t after sub-sequence t + d
	uint32_t *meta;               // per chain: steps walked | (link distance << 8) | restarted-after-garbage << 31
	unsigned long long *key01;    // [sub-sequence t][2]: chain_key of chain t after steps 0 and 1
	uint32_t *surv;               // per frame (at sub_off): the chains listed by k_huff_links
	JhSubInfo *prec;              // per sub-sequence: its record on the emitting workgroup's path (bit 63 of state: none)
	unsigned long long *wg_key;   // per chains workgroup: the entry state its emitted records assume
	JhWgAgg *wg_agg;              // per chains workgroup: totals of its emitted range, flags
	JhWgAgg *wg_base;             // per chains workgroup: exclusive prefix of the totals
	JhSubStart *start;            // entry records for the write pass (relative to the workgroup's base)
	unsigned long long *drop;     // per restart segment: (first block << 8) | blocks of the MCU the PREVIOUS segment left unfinished
	const JhDecFrame *frames;
	JhDecState *state;
	const JhHuffLut *luts;
	int16_t *coefs;
	int nframes;
	uint32_t max_chunks;          // per frame (grid size)
	uint32_t max_subs;            // per frame (grid size)
	uint32_t max_luts;            // Huffman tables of the frame that uses most
	uint32_t lut_bytes;           // LDS the tables of the frame that needs most take (DC tables in their short form), 16-aligned
	uint32_t sub_shift;           // log2 of the bits per speculative sub-sequence (one thread each): 11, or 12 for dense data
	int debug;                    // bit 0: force repairs, bit 1: force the serial walk (tests)
	int segmented;                // a frame of the batch announces restart intervals (DRI): which form of the decode loop the
	                              // synchronisation kernels take (a hint; both are right for every stream)
	JhRemap rm;
};

int jh_launch_ecs_unstuff(const JhDecodeWork &w, hipStream_t s, const JhTimerHook *th);
int jh_launch_huff_sync(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th);
int jh_launch_huff_decode(const JhDecodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th);

int jh_launch_decode_pixels(const JhGeom &g, int nframes, const int16_t *coefs, const JhQuant *quant,
                            void *out, size_t frame_stride, hipStream_t stream, float *tap = nullptr);
// What the entropy encoder wants to know about a block besides its coefficients: the map of its nonzero
// coefficients (bit k: coefficient k in zig-zag order) and one word (`acdc`) with the bits its AC symbols will
// take in the low half (write_ac, coeffs.c:164-194, EOB / ZRL included; JH_ACBITS_NOCODE: a symbol has no code,
// value_to_vlc == -1) and the quantised DC coefficient, before prediction, in the high half.  Written by the
// pixel kernel while the quantised block is still in LDS (default tables: the frame's codes are known then), or
// by k_enc_bitlen from the coefficients (optimised tables, coefficients handed in by the caller: AC bits 0);
// read by k_enc_dcbits (DC symbol + sums) and by k_enc_pack instead of a second look at the coefficients.
#define JH_ACBITS_NOCODE 0xffffu

// What the AC symbols of a block cost with a frame's codes, as stage 5 of the pixel kernel looks it up: per AC table id
// (0 / 1), zero run 0 ... 63 and category, the size of the symbol's code plus its extra bits plus (run >> 4) times
// the size of ZRL (0x4000 and up: a symbol without a code, value_to_vlc == -1); fix[id] = sizes of EOB and ZRL.
// Made once per run() from the frames' codes (k_enc_runtab) instead of by every workgroup of the pixel kernel.
struct JhRunTab {
	uint16_t run[2][64][16];
	uint16_t fix[2][2];
	uint16_t pad[4];
};

// state: NULL, or the frames' states: for P > 12 a coefficient beyond 16 bits sets JH_EST_NO_CODE there.
// codes: NULL, or the frames' codes ([frame][4]): the kernel then also fills nzmap[] / acdc[] ([frame][nblk]) with
// the AC code sizes of codes[frame * 4 + 2 + Ta], through runtab[frame] (device memory for the kernel's tables)
int jh_launch_encode_pixels(const JhGeom &g, int nframes, const void *in, size_t frame_stride,
                            const JhQuant *quant_raster, int16_t *coefs, unsigned long long *nzmap, uint32_t *acdc,
                            const JhHuffEnc *codes, JhRunTab *runtab, struct JhEncState *state, hipStream_t stream,
                            float *tap = nullptr);
// tap (both launchers): NULL, or device memory that receives the float stage boundary of the reference the kernel
// crosses -- decode: frame->data behind frame_to_rgb (frame.c:188-244), [frame][Y][X][1 or 3]; encode: the component
// planes behind transform_frame_to_components (frame.c:84-132), [frame][planes of by*8 x bx*8] -- for the stage tests

// the stages one by one (pixels.hip, bottom): a = in/out array, b = table or precision, c = out array, n = elements / blocks /
// pixels, p0.. = P (transforms) or components, X, Y, size_x (colour; P travels in b)
enum { JH_STAGE_DEQUANTIZE, JH_STAGE_QUANTIZE, JH_STAGE_IDCT, JH_STAGE_FDCT, JH_STAGE_TO_RGB, JH_STAGE_TO_YCC };
int jh_launch_stage(int what, void *a, const void *b, void *c, size_t n, int p0, int p1, int p2, size_t p3, hipStream_t s);

int jh_launch_selftest_divide(unsigned long long *d_out, int per, unsigned seed, hipStream_t stream);
int jh_launch_selftest_ycc8(unsigned long long *d_out, hipStream_t stream);

// ---- entropy encoder -------------------------------------------------------------------------------

enum {                            // JhEncState.status bits
	JH_EST_DC_RANGE = 1,          // |DC difference| > 2047 (assert in the reference, encoder.c:447)
	JH_EST_NO_CODE = 2,           // symbol missing from the table (value_to_vlc == -1)
	JH_EST_OVERFLOW = 4,          // output buffer too small
};

struct JhEncState {               // per frame, written by the device
	uint32_t status;
	uint32_t total_bits;          // entropy-coded bits before padding
	uint32_t us_bytes;            // bytes of the unstuffed stream (after padding with 1s)
	uint32_t ecs_bytes;           // bytes after 0xFF stuffing (+ RSTn markers)
	uint32_t first_err;           // ~(block * 2 + kind) of the FIRST failing block in scan order (kind 0: DC range,
	                              // 1: missing code), 0 = none: the serial encoder stops at that block with that error
};

struct JhDevSpec {                // a Huffman table specification as a DHT segment carries it (common.h:85-91)
	uint8_t L[16];
	uint8_t V[256];
	uint32_t mt;                  // number of symbols
};

struct JhEncodeWork {
	const int16_t *coefs;         // [frame][nblk][64] zig-zag
	JhHuffEnc *codes;             // [frame][Tc*2+Th]: DC0, DC1, AC0, AC1 (table ids 0/1 only, as the encoder uses)
	uint32_t *hist;               // [frame][4][257] symbol counts (optimised tables)
	uint16_t *bitlen;             // [frame][nblk]
	unsigned long long *nzmap;    // [frame][nblk] nonzero maps
	uint32_t *acdc;               // [frame][nblk] AC bits | DC coefficient << 16
	uint32_t *wg_bits;            // [frame][nwg] per-workgroup sums, then exclusive bases
	uint32_t *us;                 // unstuffed streams, frame stride us_stride bytes
	size_t us_stride;
	uint32_t *chunk_ff;           // [frame][nchunk] 0xFF counts per chunk, then exclusive bases
	uint8_t *out;                 // final codestreams: frame stride out_stride, ECS starts at hdr_len[frame]
	size_t out_stride;
	uint32_t *hdr_len;            // [frame]
	JhDevSpec *specs;             // [frame][Tc*2+Th] tables adapted on the device (optimised mode)
	const uint8_t *hdr_pre;       // header bytes in front of the DHT segments (SOI, DQT, SOF0), same for every frame
	const uint8_t *hdr_suf;       // header bytes behind them ([DRI], SOS)
	int hdr_pre_len, hdr_suf_len;
	int ntab;                     // table ids in use: 1 (one component) or 2
	JhEncState *state;
	int nframes;
	int nwg;                      // workgroups (256 blocks each) per frame
	int nchunk;                   // stuffing chunks per frame (capacity)
	int restart_interval;         // MCUs, 0 = none (EXTENSION: the reference never writes DRI / RSTn)
	int nivl;                     // restart intervals per frame (restart_interval > 0)
	uint32_t *ivl_raw;            // [frame][nivl] entropy-coded bits up to the END of interval k, before padding
	uint32_t *ivl_padsum;         // [frame][nivl + 1] padding bits (to byte boundaries) before interval k
	uint32_t *ivl_eb;             // [frame][nivl] byte offset in the unstuffed stream at which interval k ends
};

int jh_launch_enc_histogram(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th);
int jh_launch_enc_adapt(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th);
int jh_launch_enc_headers(const JhEncodeWork &w, const JhGeom &g, hipStream_t s);
// acbits_ready: the pixel kernel has filled w.nzmap / w.acdc with the codes in w.codes: the bit lengths then need
// the DC symbols only (k_enc_dcbits); otherwise k_enc_bitlen walks the coefficients and fills both arrays
int jh_launch_enc_entropy(const JhEncodeWork &w, const JhGeom &g, hipStream_t s, const JhTimerHook *th, bool acbits_ready);
// Motion JPEG stream out: the finished codestreams of the batch back to back at dst (device memory, or page-locked
// host memory the device can write), at most cap bytes; off[nframes + 1] / len[nframes] (device memory) receive
// where each frame went and how long it is (0: the frame failed, JH_STREAM_NOFIT: no room left for it)
#define JH_STREAM_NOFIT 0xffffffffu
int jh_launch_enc_stream(const JhEncodeWork &w, void *dst, size_t cap, unsigned long long *off, uint32_t *len,
                         hipStream_t s, const JhTimerHook *th);
