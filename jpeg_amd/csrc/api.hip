// jpeg_amd/csrc/api.hip -- the C-ABI of libjpeg_amd.so (declared in include/jpeg_amd.h).
//
// Host orchestration only: header parsing / production (codestream.cpp), workspace management and
// kernel launches.  Everything pixel- or coefficient-sized stays resident in HBM between stages.
// There is no CPU fallback: without a usable HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <system_error>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <vector>
#include <algorithm>
#include <map>
#include <thread>
#include <utility>
#include <chrono>
#include <string>
#include <functional>
#include <mutex>
#include <atomic>
#include "../../include/jpeg_amd.h"
#include "common.hpp"
#include "codestream.hpp"
#include "kernels.hpp"

#define JH_MAX_PASSES 32              // scans per frame the decoder takes (each is a pass of the Huffman kernels with
                                      // buffers of its own; the reference takes any number, decoder.c:593-601)

#define HIPCHK(call)                                                                         \
	do {                                                                                     \
		hipError_t e_ = (call);                                                              \
		if (e_ != hipSuccess) {                                                              \
			fprintf(stderr, "jpeg_amd: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), \
			        __FILE__, __LINE__);                                                     \
			return e_ == hipErrorOutOfMemory ? RET_FAILURE_MEMORY_ALLOCATION : RET_FAILURE_LOGIC_ERROR; \
		}                                                                                    \
	} while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int jpeg_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		return 0;
	}
	return n;
}

// ABI 4 (round 3): jpeg_hip_decoder_set_lanes added.  ABI 3 (round 3): jpeg_hip_encoder_run_stream / sync_stream, jpeg_hip_streams_create / destroy, the float taps,
// jpeg_hip_selftest_ycc8 added; nothing removed.  ABI 2 (round 2): jpeg_hip_params lost its restart_interval member
// (now jpeg_hip_encoder_set_restart_interval) -- a caller built against the older header must be rebuilt.
extern "C" const char *jpeg_hip_version(void) { return "jpeg_amd 0.3 (gfx950, ABI 4)"; }
extern "C" int jpeg_hip_abi_version(void) { return JPEG_HIP_ABI_VERSION; }

// host only (no device needed): frames of a Motion JPEG stream
extern "C" int jpeg_hip_mjpeg_split(const uint8_t *stream, size_t len, size_t *frame_off, size_t *frame_len, int cap)
{
	if (!stream || cap < 0 || (cap > 0 && (!frame_off || !frame_len))) {
		return -1;
	}
	return jh_split_stream(stream, len, frame_off, frame_len, cap);
}

static int require_device(int device)
{
	int n = jpeg_hip_device_count();
	if (n <= 0) {
		fprintf(stderr, "jpeg_amd: no HIP device available -- this library has no CPU fallback\n");
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (device < 0 || device >= n) {
		fprintf(stderr, "jpeg_amd: device %d out of range (%d devices)\n", device, n);
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(device));
	return RET_SUCCESS;
}

static void fill_info(const JhGeom &g, jpeg_hip_frame_info *o)
{
	memset(o, 0, sizeof *o);
	o->P = g.P; o->X = g.X; o->Y = g.Y; o->Nf = g.Nf;
	for (int k = 0; k < g.ncomp; ++k) {
		o->id[k] = g.id[k]; o->H[k] = g.H[k]; o->V[k] = g.V[k];
		o->Tq[k] = g.Tq[k]; o->Td[k] = g.Td[k]; o->Ta[k] = g.Ta[k];
		o->bx[k] = g.bx[k]; o->by[k] = g.by[k];
	}
	o->max_H = g.max_H; o->max_V = g.max_V; o->m_x = g.m_x; o->m_y = g.m_y;
	o->size_x = g.size_x; o->size_y = g.size_y; o->Ri = g.Ri;
	o->blocks_per_mcu = g.B; o->out_components = g.out_comps; o->sample_bytes = g.sample_bytes;
}

extern "C" int jpeg_hip_parse_info(const uint8_t *jpg, size_t len, int tolerant, jpeg_hip_frame_info *info)
{
	if (!jpg || !info) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	JhHeader h;
	int err = jh_parse_headers(jpg, len, &h, tolerant != 0);
	if (!err) {
		for (int j = 0; j < h.g.ncomp; ++j) {
			h.g.scan_order[j] = h.Ns == h.g.ncomp ? h.scan_comp[j] : j;   // Ns == 0: no scan before EOI
		}
		err = jh_finish_geometry(&h.g);
	}
	if (!err) {
		fill_info(h.g, info);
	}
	return err;
}

// scan-order / zig-zag int16 coefficients of one frame -> the reference's int_buffer layout of component k
static void coefs_to_reference_layout(const JhGeom &g, const int16_t *fc, int k, int32_t *out)
{
	for (int by = 0; by < g.by[k]; ++by) {
		for (int bx = 0; bx < g.bx[k]; ++bx) {
			const int mx = bx / g.H[k], h = bx % g.H[k], my = by / g.V[k], v = by % g.V[k];
			const size_t blk = (size_t)(my * g.m_x + mx) * g.B + g.comp_first[k] + v * g.H[k] + h;
			int32_t *dst = out + ((size_t)by * g.bx[k] + bx) * 64;
			for (int z = 0; z < 64; ++z) {
				dst[jh_zigzag[z]] = fc[blk * 64 + z];
			}
		}
	}
}

struct StageTimer {
	bool on = false;
	std::vector<std::string> names;
	std::vector<hipEvent_t> ev;       // 2 per stage
	size_t used = 0;
	void begin(const char *name, hipStream_t s)
	{
		if (!on) {
			return;
		}
		if (used + 2 > ev.size()) {
			hipEvent_t a, b;
			hipEventCreate(&a);
			hipEventCreate(&b);
			ev.push_back(a);
			ev.push_back(b);
		}
		names.push_back(name);
		hipEventRecord(ev[used], s);
	}
	void end(hipStream_t s)
	{
		if (!on) {
			return;
		}
		hipEventRecord(ev[used + 1], s);
		used += 2;
	}
	void reset() { used = 0; names.clear(); }
	int collect(const char **n, float *ms, int cap)
	{
		int k = 0;
		for (size_t i = 0; i + 1 < used && k < cap && (size_t)k < names.size(); i += 2, ++k) {
			hipEventSynchronize(ev[i + 1]);
			float t = 0.f;
			hipEventElapsedTime(&t, ev[i], ev[i + 1]);
			n[k] = names[k].c_str();
			ms[k] = t;
		}
		return k;
	}
	static void hook_begin(void *c, const char *n, hipStream_t s) { ((StageTimer *)c)->begin(n, s); }
	static void hook_end(void *c, hipStream_t s) { ((StageTimer *)c)->end(s); }
	JhTimerHook hook() { return JhTimerHook{hook_begin, hook_end, on ? this : nullptr}; }
	~StageTimer()
	{
		for (hipEvent_t e : ev) {
			hipEventDestroy(e);
		}
	}
};

// ======================================================================================= decoder

// One scan of every frame of the batch: its own unstuffed streams, synchronisation records and states.
// Frames whose single scan interleaves all components have one pass; frames coded as one scan per
// component (non-interleaved, decoder.c:274-302) have one pass per scan.  The raw codestreams, the
// quantisation tables and the coefficient buffer are shared.
struct DecPass {
	JhGeom g;                         // what the Huffman kernels see (B = 1 for a single-component scan)
	int comp = -1;                    // component of a single-component scan, -1 = all (interleaved), -2 = some
	std::vector<JhDecFrame> frames;
	std::vector<JhDecState> state;
	uint8_t *d_cs = nullptr;
	uint32_t *d_seg = nullptr, *d_chunk_kept = nullptr, *d_chunk_rst = nullptr;
	JhSubInfo *d_rec = nullptr, *d_prec = nullptr;
	unsigned long long *d_wg_key = nullptr;
	uint32_t *d_meta = nullptr, *d_surv = nullptr;
	unsigned long long *d_key01 = nullptr;
	JhWgAgg *d_wg_agg = nullptr, *d_wg_base = nullptr;
	JhSubStart *d_start = nullptr;
	unsigned long long *d_drop = nullptr;
	JhDecFrame *d_frames = nullptr;
	JhDecState *d_state = nullptr;
	JhDecState *h_state = nullptr;    // page-locked: run() ends with a copy of the states into it, on its stream
	size_t h_state_cap = 0;
	JhHuffLut *d_luts = nullptr;
	std::vector<JhHuffLut> luts;
	size_t cs_off = 0, seg_off = 0, chunk_off = 0, sub_off = 0, wg_off = 0;
	uint32_t max_chunks = 1, max_subs = 1, max_luts = 1, lut_bytes = sizeof(JhHuffLut);
	int segmented = 0;                // a frame's scan came with a restart interval (DRI)
	uint32_t sub_shift = JH_SUBSEQ_SHIFT; // log2 of the bits per sub-sequence of this pass
	JhDecodeWork work{};

	void release()
	{
		hipFree(d_cs); hipFree(d_seg); hipFree(d_chunk_kept); hipFree(d_chunk_rst);
		hipFree(d_start); hipFree(d_frames); hipFree(d_state); hipFree(d_luts);
		hipFree(d_rec); hipFree(d_meta); hipFree(d_surv); hipFree(d_key01); hipFree(d_wg_agg); hipFree(d_wg_base); hipFree(d_prec); hipFree(d_wg_key); hipFree(d_drop);
		d_drop = nullptr;
		if (h_state) {
			hipHostFree(h_state);
		}
		h_state = nullptr;
		h_state_cap = 0;
		d_cs = nullptr; d_seg = d_chunk_kept = d_chunk_rst = nullptr;
		d_start = nullptr; d_frames = nullptr; d_state = nullptr; d_luts = nullptr;
		d_rec = d_prec = nullptr; d_meta = nullptr; d_surv = nullptr; d_key01 = nullptr; d_wg_agg = d_wg_base = nullptr; d_wg_key = nullptr;
	}
};

// Threads that fill the staging buffer of a prepare() from pageable memory (this one included): at most 6, and no more
// than this process's share of the host -- the CPUs it may run on (sched_getaffinity) divided by the ranks that
// share the node (LOCAL_WORLD_SIZE / WORLD_SIZE of torch.distributed.run; one process per GPU), less one for the
// caller's own thread.  8 ranks on a 16-core share: 1 thread each beside Python's, instead of 8 x 6 = 48.
// JPEG_AMD_FILL_THREADS overrides.
static int fill_threads()
{
	static const int n = [] {
		if (const char *e = getenv("JPEG_AMD_FILL_THREADS")) {
			const int v = atoi(e);
			if (v >= 1 && v <= 64) {
				return v;
			}
		}
		int cpus = (int)std::thread::hardware_concurrency();
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof set, &set) == 0) {
			cpus = CPU_COUNT(&set);
		}
		int ranks = 1;
		for (const char *name : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"}) {
			if (const char *e = getenv(name)) {
				ranks = atoi(e) > 0 ? atoi(e) : 1;
				break;
			}
		}
		const int share = cpus / ranks - 1;
		return share < 1 ? 1 : (share > 6 ? 6 : share);
	}();
	return n;
}

#define JH_MAX_LANES 4                 // one per hardware queue of the process
#define JH_UPLOAD_PIECE (8u << 20)     // bytes per DMA of a batch that is uploaded from caller-pinned memory

struct jpeg_hip_decoder {
	int device = 0;
	int nframes = 0;
	float *float_tap = nullptr;               // stage tests: see jpeg_hip_decoder_set_float_tap()
	JhGeom g;
	std::vector<DecPass> pass;
	std::vector<int> parse_err;
	std::vector<size_t> ecs_len;
	std::vector<size_t> file_off, file_len;   // the frames inside d_raw (valid until the next prepare)
	std::vector<const uint8_t *> file_ptr;    // and on the host: inside h_stage, the caller's pinned memory, or patched[]
	std::vector<std::vector<uint8_t>> patched;    // private copies of frames sync() had to cut (caller's memory is never written)
	std::vector<size_t> spliced;              // bytes sync() cut out of a frame behind stray markers (see there)
	// device
	uint8_t *d_raw = nullptr;
	uint8_t *h_stage = nullptr;       // pinned staging for the codestreams, kept between prepare() calls
	size_t stage_bytes = 0;
	JhQuant *d_quant = nullptr;
	int16_t *d_coefs = nullptr;
	size_t raw_bytes = 0;
	void *last_out = nullptr;
	size_t last_stride = 0;
	hipStream_t last_stream = nullptr;
	hipStream_t copy_stream = nullptr;    // the codestreams' DMA runs here, beside the kernels of another decoder
	                                      // (made by the first prepare() that wants it: a stream that exists takes
	                                      // one of the process's few hardware queues, used or not)
	hipStream_t up_stream = nullptr;      // ... or on this stream of the caller's (jpeg_hip_decoder_set_upload_stream)
	bool up_user = false;
	hipEvent_t uploaded = nullptr;        // ... and run() makes its stream wait for this
	hipEvent_t done = nullptr;            // end of run()'s kernels: sync() waits for it, not for the whole stream
	// jpeg_hip_decoder_set_lanes(): run() lays the batch's frames out over `lanes` streams -- the caller's and these --
	// so that the narrow kernels of one part (scans, links, survivors: a few waves per SIMD, waiting on each other)
	// run beside the wide ones of the others
	int lanes = 1;
	bool lanes_chosen = false;            // by a call; otherwise JPEG_AMD_LANES (1 ... 4) applies at the first run()
	bool lanes_owned = false;             // the lane streams are this object's (not handed in by the caller)
	hipStream_t lane_stream[JH_MAX_LANES - 1] = {nullptr, nullptr, nullptr};
	hipEvent_t lane_fork = nullptr, lane_done[JH_MAX_LANES - 1] = {nullptr, nullptr, nullptr};
	bool ran = false;
	int debug = 0;
	bool tolerant = false;
	bool coefs_only = false;          // run() stops behind the Huffman passes (jpeg_hip_read_scan)
	bool all_scans_tried = false;     // sync() has prepared this batch again with every scan found on the host
	StageTimer timer;

	// Device buffers only grow: a decoder that is prepared again with a batch of similar size (Motion
	// JPEG) allocates nothing, and hipMalloc / hipFree would wait for every stream of the device.
	std::map<void **, size_t> cap;    // keyed by the address of the pointer member (pass never relocates)
	hipError_t grow(void **p, size_t bytes)
	{
		size_t &c = cap[p];
		if (*p && c >= bytes) {
			return hipSuccess;
		}
		if (*p) {
			hipFree(*p);
			*p = nullptr;
			c = 0;
		}
		const size_t want = bytes + bytes / 4 + 256;
		hipError_t e = hipMalloc(p, want);
		if (e == hipSuccess) {
			c = want;
		}
		return e;
	}
	void release()
	{
		for (DecPass &p : pass) {
			p.release();
		}
		pass.clear();
		hipFree(d_raw); hipFree(d_quant); hipFree(d_coefs);
		d_raw = nullptr; d_quant = nullptr; d_coefs = nullptr;
		cap.clear();
	}
};

extern "C" int jpeg_hip_decoder_create(jpeg_hip_decoder **dec, int device)
{
	*dec = nullptr;
	int err = require_device(device);
	if (err) {
		return err;
	}
	jpeg_hip_decoder *d = new jpeg_hip_decoder();
	d->device = device;
	d->pass.reserve(JH_MAX_PASSES);                        // addresses of the passes stay put (see cap)
	memset(&d->g, 0, sizeof d->g);
	if (hipSetDevice(device) != hipSuccess ||
	    hipEventCreateWithFlags(&d->uploaded, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&d->done, hipEventDisableTiming) != hipSuccess) {
		jpeg_hip_decoder_destroy(d);
		return RET_FAILURE_MEMORY_ALLOCATION;
	}
	*dec = d;
	return RET_SUCCESS;
}

static void decoder_drop_lanes(jpeg_hip_decoder *dec)
{
	for (int k = 0; k < JH_MAX_LANES - 1; ++k) {
		if (dec->lane_stream[k]) {
			hipStreamSynchronize(dec->lane_stream[k]);
			if (dec->lanes_owned) {
				hipStreamDestroy(dec->lane_stream[k]);
			}
			dec->lane_stream[k] = nullptr;
		}
		if (dec->lane_done[k]) {
			hipEventDestroy(dec->lane_done[k]);
			dec->lane_done[k] = nullptr;
		}
	}
	if (dec->lane_fork) {
		hipEventDestroy(dec->lane_fork);
		dec->lane_fork = nullptr;
	}
	dec->lanes = 1;
}

extern "C" void jpeg_hip_decoder_destroy(jpeg_hip_decoder *dec)
{
	if (dec) {
		hipSetDevice(dec->device);
		if (dec->copy_stream) {
			hipStreamSynchronize(dec->copy_stream);
		}
		dec->release();
		if (dec->h_stage) {
			hipHostFree(dec->h_stage);
		}
		if (dec->uploaded) {
			hipEventDestroy(dec->uploaded);
		}
		if (dec->done) {
			hipEventDestroy(dec->done);
		}
		if (dec->copy_stream) {
			hipStreamDestroy(dec->copy_stream);
		}
		decoder_drop_lanes(dec);
		delete dec;
	}
}

static bool same_geometry(const JhGeom &a, const JhGeom &b)
{
	if (a.P != b.P || a.X != b.X || a.Y != b.Y || a.Nf != b.Nf || a.ncomp != b.ncomp) {
		return false;
	}
	for (int k = 0; k < a.ncomp; ++k) {
		if (a.H[k] != b.H[k] || a.V[k] != b.V[k] || a.scan_order[k] != b.scan_order[k]) {
			return false;
		}
	}
	return true;
}

// one scan of one frame as the host found it
struct ScanDesc {
	int comp;                         // -1: all components interleaved, -2: some of them (cs[0 .. ncs)), else the one
	int ncs, cs[JH_MAX_COMP];         // the scan's components in the order of its header
	size_t ecs_begin, ecs_end;        // ecs_end: end of the bytes the device may look at
	int Td[JH_MAX_COMP], Ta[JH_MAX_COMP];
	JhHuffSpec ht[2][4];              // the tables in force when the scan started
	int Ri;
};

// SOI .. EOI of one frame: headers, and the list of scans.  Tables may be redefined between scans
// (decoder.c:581-591 runs whenever a DHT comes); quantisation tables count as they stand at EOI,
// where the reference dequantises (decoder.c:456-470).
// cut (optional): a scan of some components that ends on a marker behind which only skippable segments and then
// RSTn follow goes on at that RSTn in the reference (decoder.c:622-633, see jpeg_hip_decoder_sync).  Then the
// bytes [cut[0], cut[1]) are the ones to take out and PARSE_SPLICE is returned: the caller parses the file
// without them.
enum { PARSE_SPLICE = 0x7fff0001 };

static int parse_frame(const uint8_t *p, size_t n, JhHeader *h, std::vector<ScanDesc> *scans, bool tolerant, size_t *cut = nullptr,
                       bool all_scans = false)
{
	scans->clear();
	int err = jh_parse_headers(p, n, h, tolerant);
	if (err) {
		return err;
	}
	auto describe = [&](int comp, size_t end) {
		ScanDesc sd;
		sd.comp = comp;
		sd.ncs = h->Ns;
		for (int j = 0; j < JH_MAX_COMP; ++j) {
			sd.cs[j] = j < h->Ns ? h->scan_comp[j] : -1;
		}
		sd.ecs_begin = h->ecs_offset;
		sd.ecs_end = end;
		for (int k = 0; k < JH_MAX_COMP; ++k) {
			sd.Td[k] = h->g.Td[k];
			sd.Ta[k] = h->g.Ta[k];
		}
		memcpy(sd.ht, h->ht, sizeof sd.ht);
		sd.Ri = h->g.Ri;
		scans->push_back(sd);
	};
	if (h->Ns == 0) {
		// no scan at all: the frame of zero coefficients (mid grey), as the reference writes it
		for (int j = 0; j < h->g.ncomp; ++j) {
			h->g.scan_order[j] = j;
		}
		return RET_SUCCESS;
	}
	if (h->Ns == h->g.ncomp && !all_scans) {
		// One scan with every component: the device finds its end (no host pass over the data).  Should another
		// scan follow it, sync() sees that and prepares the batch once more with all_scans set.
		for (int j = 0; j < h->Ns; ++j) {
			h->g.scan_order[j] = h->scan_comp[j];
		}
		describe(-1, n);
		return RET_SUCCESS;
	}
	// Several scans, each with one component, some of them, or all: every scan is a pass over its own data in the
	// file's order, and a component that is coded again is written again (read_block() stores every coefficient
	// of a block it reads, coeffs.c:222-229; decoder.c:593-601 asks no questions).
	for (int nscan = 0;; ++nscan) {
		if (nscan >= JH_MAX_PASSES) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		const int k = h->Ns == 1 ? h->scan_comp[0] : -2;
		const size_t end = jh_ecs_end(p, n, h->ecs_offset);
		if (end >= n) {
			return RET_FAILURE_FILE_IO;                    // no marker after the data (io.c:197)
		}
		size_t rst = 0;
		if (cut && p[end + 1] != 0xd9 && jh_walk_to_restart(p, n, end, tolerant, &rst) == RET_SUCCESS && rst > end) {
			cut[0] = end;
			cut[1] = rst;
			return PARSE_SPLICE;
		}
		describe(k, end + 2);
		bool eoi = false;
		err = jh_parse_next_scan(p, n, end, h, &eoi);
		if (err) {
			return err;
		}
		if (eoi) {
			break;
		}
	}
	for (int j = 0; j < h->g.ncomp; ++j) {
		h->g.scan_order[j] = j;                            // buffer order: components in frame order
	}
	return RET_SUCCESS;
}

// headers and scans of frame f: parse_frame() for codestreams, or the caller's description (jpeg_hip_read_scan)
typedef std::function<int(int, JhHeader *, std::vector<ScanDesc> *)> FrameParser;

// pinned (optional, in/out): the frames lie in page-locked host memory, close together (a Motion JPEG file read
// into one pinned buffer): then the span that holds them goes to the device in ONE DMA from where it lies, and the
// frames keep their distances there (the kernels take any offset: their 4 KiB chunks are cut by every frame's
// own bounds).  The parser may clear the flag; a span that is mostly holes clears it too.
static int decoder_prepare(jpeg_hip_decoder *dec, int nframes, const uint8_t *const *jpg, const size_t *jpg_len,
                           const FrameParser &parse, bool *pinned = nullptr)
{
	if (!dec || nframes <= 0) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(dec->device));
	if (dec->ran) {
		HIPCHK(hipEventSynchronize(dec->done));            // the buffers are about to be reused
	}
	HIPCHK(hipEventSynchronize(dec->uploaded));            // ... and the staging buffer (prepare() twice in a row)
	dec->ran = false;
	dec->nframes = 0;                                      // not prepared until everything below has succeeded
	dec->parse_err.assign(nframes, 0);
	dec->ecs_len.assign(nframes, 0);
	dec->spliced.assign(nframes, 0);

	auto T0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (getenv("JPEG_HIP_TRACE")) {
			auto T1 = std::chrono::steady_clock::now();
			fprintf(stderr, "[prepare] %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(T1 - T0).count());
			T0 = T1;
		}
	};
	std::vector<JhHeader> hdr(nframes);
	std::vector<std::vector<ScanDesc>> scans(nframes);
	std::vector<JhQuant> quant(nframes);
	std::vector<size_t> &file_off = dec->file_off;
	file_off.assign(nframes, 0);
	std::vector<ScanDesc> pass_comp;                       // the batch's scan structure (from the first good frame)
	bool have_geom = false;
	size_t raw_off = 0;

	for (int f = 0; f < nframes; ++f) {
		JhHeader &h = hdr[f];
		int err = parse(f, &h, &scans[f]);
		if (!err) {
			err = jh_finish_geometry(&h.g);
		}
		if (!err) {
			if (!have_geom) {
				dec->g = h.g;
				have_geom = true;
				for (const ScanDesc &sd : scans[f]) {
					pass_comp.push_back(sd);
				}
			} else {
				// one batch = one geometry and one scan structure
				bool same = same_geometry(dec->g, h.g) && scans[f].size() == pass_comp.size();
				for (size_t i = 0; same && i < pass_comp.size(); ++i) {
					same = scans[f][i].comp == pass_comp[i].comp && scans[f][i].ncs == pass_comp[i].ncs &&
					       memcmp(scans[f][i].cs, pass_comp[i].cs, sizeof pass_comp[i].cs) == 0;
				}
				if (!same) {
					err = RET_FAILURE_FILE_UNSUPPORTED;
				}
			}
		}
		dec->parse_err[f] = err;
		if (!err) {
			for (int t = 0; t < 4; ++t) {
				for (int z = 0; z < 64; ++z) {
					quant[f].q[t][z] = h.Q[t][jh_zigzag[z]];
				}
			}
		}
	}
	// where the frames go in d_raw (after the parsing: it may hand back a shorter file, in other memory)
	dec->file_len.assign(jpg_len, jpg_len + nframes);
	dec->file_ptr.assign(nframes, nullptr);
	dec->patched.clear();
	dec->patched.resize(nframes);
	const uint8_t *span_lo = nullptr, *span_hi = nullptr;
	bool direct = pinned && *pinned;
	if (direct) {
		size_t sum = 0;
		for (int f = 0; f < nframes; ++f) {
			span_lo = !span_lo || jpg[f] < span_lo ? jpg[f] : span_lo;
			span_hi = !span_hi || jpg[f] + jpg_len[f] > span_hi ? jpg[f] + jpg_len[f] : span_hi;
			sum += jpg_len[f];
		}
		const size_t span = (size_t)(span_hi - span_lo);
		direct = span <= 2 * sum + (1u << 20) && span + 8192 <= 0xf0000000ull;
	}
	if (pinned) {
		*pinned = direct;
	}
	const size_t lead = direct ? (size_t)((uintptr_t)span_lo & 4095) : 0;      // the span keeps its place in its page
	for (int f = 0; f < nframes; ++f) {
		if (direct) {
			file_off[f] = lead + (size_t)(jpg[f] - span_lo);
			continue;
		}
		file_off[f] = raw_off;
		if (!dec->parse_err[f]) {
			raw_off = align_up(raw_off + jpg_len[f] + 32, 4096);   // a frame starts a 4 KiB chunk of the un-stuffing kernels
			if (raw_off > 0xf0000000ull) {
				return RET_FAILURE_OVERFLOW_ERROR;               // 32-bit offsets: split the batch
			}
		}
	}
	if (direct) {
		raw_off = align_up(lead + (size_t)(span_hi - span_lo) + 32, 4096);
	}
	lap("parse");
	if (!have_geom) {
		// nothing decodable: report the first parse error
		return dec->parse_err[0] ? dec->parse_err[0] : RET_FAILURE_FILE_UNSUPPORTED;
	}
	if (raw_off == 0) {
		raw_off = 4096;
	}
	dec->raw_bytes = raw_off + 4096;
	const JhGeom &G = dec->g;

	if (pass_comp.size() > (size_t)JH_MAX_PASSES) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	while (dec->pass.size() > pass_comp.size()) {
		dec->pass.back().release();
		dec->pass.pop_back();
	}
	// Everything prepare() puts on the device goes through ONE stream -- the caller's (set_upload_stream) or the
	// decoder's copy stream -- and nothing through the null stream: that one shares a hardware queue with some
	// other stream of the process, and a blocking call on it waits for whatever that stream has queued.
	if (!dec->up_user && !dec->copy_stream) {
		HIPCHK(hipStreamCreateWithFlags(&dec->copy_stream, hipStreamNonBlocking));
	}
	const hipStream_t ups = dec->up_user ? dec->up_stream : dec->copy_stream;
	dec->pass.resize(pass_comp.size());
	for (size_t pi = 0; pi < pass_comp.size(); ++pi) {
		DecPass &ps = dec->pass[pi];
		ps.comp = pass_comp[pi].comp;
		const int ncs = pass_comp[pi].ncs;
		const int *cs = pass_comp[pi].cs;
		ps.g = G;
		ps.luts.clear();
		ps.cs_off = ps.seg_off = ps.chunk_off = ps.sub_off = ps.wg_off = 0;
		ps.segmented = 0;
		ps.max_chunks = ps.max_subs = ps.max_luts = 1;
		ps.lut_bytes = sizeof(JhHuffLut);
		JhRemap rm;
		memset(&rm, 0, sizeof rm);
		rm.nblk_full = G.nblk;
		if (ps.comp >= 0) {
			const int k = ps.comp;
			// the reference's "MCU" of such a scan is H x V consecutive raster blocks (decoder.c:283-289):
			// that is the unit dropped when an interval ends inside it, so it is the unit here as well
			ps.g.B = G.H[k] * G.V[k];
			for (int b = 0; b < ps.g.B; ++b) {
				ps.g.blk_comp[b] = 0;
			}
			ps.g.nblk = G.bx[k] * G.by[k];
			ps.g.nmcu = ps.g.nblk / ps.g.B;
			rm.mode = 1;
			rm.bxc = G.bx[k];
			rm.H = G.H[k];
			rm.V = G.V[k];
			rm.m_x = G.m_x;
			rm.B = G.B;
			rm.kfirst = G.comp_first[k];
		} else if (ps.comp == -2) {
			// some of the components: the frame's MCUs with only their blocks, in the scan header's order
			// (decoder.c:319-358)
			rm.mode = 2;
			rm.B = G.B;
			int nk = 0;
			for (int j = 0; j < ncs; ++j) {
				const int k = cs[j];
				for (int i = 0; i < G.H[k] * G.V[k]; ++i) {
					ps.g.blk_comp[nk] = (uint8_t)j;            // index of the DC predictor
					rm.kmap[nk++] = (uint8_t)(G.comp_first[k] + i);
				}
			}
			rm.nk = nk;
			ps.g.B = nk;
			ps.g.nblk = G.nmcu * nk;
		}
		// Bits per sub-sequence.  How far a chain runs before it falls in with the true parse goes with the symbols,
		// so in bits with the bits per block: 2048 is the measured best for ordinary images (60 - 150 bits per
		// block), dense data (8K 12-bit q95: 420) does better with 4096 (70 -> 80 Gpix/s).
		{
			size_t bytes = 0, frames_ok = 0;
			for (int f = 0; f < nframes; ++f) {
				if (!dec->parse_err[f]) {
					bytes += scans[f][pi].ecs_end - scans[f][pi].ecs_begin;
					frames_ok++;
				}
			}
			const size_t blocks = frames_ok * (size_t)(ps.g.nblk > 0 ? ps.g.nblk : 1);
			ps.sub_shift = blocks && bytes * 8 / blocks >= 300 ? JH_SUBSEQ_SHIFT_DENSE : JH_SUBSEQ_SHIFT;
		}
		ps.frames.assign(nframes, JhDecFrame());
		ps.state.assign(nframes, JhDecState());
		std::vector<std::pair<int, JhHuffSpec>> prev_specs;
		uint32_t prev_lut_off = 0;
		for (int f = 0; f < nframes; ++f) {
			JhDecFrame &fr = ps.frames[f];
			memset(&fr, 0, sizeof fr);
			int err = dec->parse_err[f];
			if (err) {
				// keep a harmless empty frame so that the grids stay rectangular
				fr.raw_begin = fr.raw_end = (uint32_t)file_off[f];
				fr.cs_off = (uint32_t)ps.cs_off;
				fr.sub_off = (uint32_t)ps.sub_off;
				fr.sub_cap = 1;
				fr.seg_off = (uint32_t)ps.seg_off;
				fr.seg_cap = 4;
				fr.chunk_off = (uint32_t)ps.chunk_off;
				fr.nchunk = 0;
				fr.lut_off = (uint32_t)ps.luts.size();
				fr.n_lut = 0;
				fr.wg_off = (uint32_t)ps.wg_off;
				ps.wg_off += 2;
				ps.cs_off += 64;
				ps.sub_off += 1;
				ps.seg_off += 4;
				continue;
			}
			const ScanDesc &sd = scans[f][pi];
			ps.segmented |= sd.Ri > 0;
			fr.raw_begin = (uint32_t)(file_off[f] + sd.ecs_begin);
			fr.raw_end = (uint32_t)(file_off[f] + sd.ecs_end);
			const size_t ecs = sd.ecs_end - sd.ecs_begin;
			if (ecs > 0x1ffffff0ull) {
				return RET_FAILURE_OVERFLOW_ERROR;               // bit positions are 32-bit: < 512 MB of entropy-coded data per scan
			}
			fr.cs_off = (uint32_t)ps.cs_off;
			const size_t cs_cap = align_up(ecs + 32 + 1024, 64);     // the stream readers look some words past the end
			fr.sub_off = (uint32_t)ps.sub_off;
			fr.sub_cap = (uint32_t)(cs_cap / ((size_t)1 << (ps.sub_shift - 3)) + 2);
			fr.seg_off = (uint32_t)ps.seg_off;
			// a single-component scan's interval counts blocks by the standard (the reference ignores Ri)
			fr.seg_cap = (uint32_t)((sd.Ri > 0 ? ((ps.comp >= 0 ? ps.g.nblk : ps.g.nmcu) + sd.Ri - 1) / sd.Ri : 1) + 64);
			fr.chunk_off = (uint32_t)ps.chunk_off;
			fr.wg_off = (uint32_t)ps.wg_off;
			ps.wg_off += fr.sub_cap / JH_SEQ_OWN + 2;
			fr.nchunk = (uint32_t)((fr.raw_end + 4095) / 4096 - fr.raw_begin / 4096);
			// Huffman tables used by this scan -> LDS slots.  Frames of a Motion JPEG stream mostly share
			// their tables: a frame whose tables equal the previous frame's reuses its look-up tables.
			int slot_of[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
			std::vector<std::pair<int, const JhHuffSpec *>> used;
			for (int j = 0; j < (ps.comp == -1 ? G.ncomp : ncs); ++j) {
				const int k = ps.comp == -1 ? j : cs[j];
				const int ids[2] = {sd.Td[k], sd.Ta[k]};
				for (int tc = 0; tc < 2; ++tc) {
					if (slot_of[tc][ids[tc]] < 0) {
						slot_of[tc][ids[tc]] = (int)used.size();
						used.push_back(std::make_pair(tc, &sd.ht[tc][ids[tc]]));
					}
				}
			}
			bool same_as_prev = prev_specs.size() == used.size();
			for (size_t i = 0; same_as_prev && i < used.size(); ++i) {
				same_as_prev = prev_specs[i].first == used[i].first &&
				               memcmp(&prev_specs[i].second, used[i].second, sizeof(JhHuffSpec)) == 0;
			}
			if (same_as_prev) {
				fr.lut_off = prev_lut_off;
				fr.n_lut = (uint32_t)used.size();
			} else {
				fr.lut_off = (uint32_t)ps.luts.size();
				for (size_t i = 0; i < used.size() && !err; ++i) {
					JhHuffLut l;
					err = jh_build_declut(used[i].second, used[i].first, &l);
					fr.n_lut++;
					ps.luts.push_back(l);
				}
				prev_specs.clear();
				if (!err) {
					for (size_t i = 0; i < used.size(); ++i) {
						prev_specs.push_back(std::make_pair(used[i].first, *used[i].second));
					}
					prev_lut_off = fr.lut_off;
				}
			}
			if (err) {
				dec->parse_err[f] = err;
				fr.nchunk = 0;
				fr.raw_end = fr.raw_begin;
			}
			for (int b = 0; b < ps.g.B; ++b) {
				const int k = ps.comp == -1 ? G.blk_comp[b] : cs[ps.g.blk_comp[b]];
				fr.blk_dc[b] = (uint8_t)slot_of[0][sd.Td[k]];
				fr.blk_ac[b] = (uint8_t)slot_of[1][sd.Ta[k]];
			}
			ps.max_chunks = fr.nchunk > ps.max_chunks ? fr.nchunk : ps.max_chunks;
			ps.max_subs = fr.sub_cap > ps.max_subs ? fr.sub_cap : ps.max_subs;
			ps.max_luts = fr.n_lut > ps.max_luts ? fr.n_lut : ps.max_luts;
			{
				// what load_tables() (huffcore.hpp) lays out in LDS: a slot that some block uses as its DC table is short
				uint32_t bytes = 0;
				for (uint32_t sl = 0; sl < fr.n_lut; ++sl) {
					bool isdc = false;
					for (int b = 0; b < ps.g.B; ++b) {
						isdc = isdc || fr.blk_dc[b] == sl;
					}
					bytes += isdc ? JH_LUT_DC_BYTES : (uint32_t)sizeof(JhHuffLut);
				}
				ps.lut_bytes = bytes > ps.lut_bytes ? bytes : ps.lut_bytes;
			}
			ps.cs_off += cs_cap;
			ps.sub_off += fr.sub_cap;
			ps.seg_off += fr.seg_cap;
			ps.chunk_off += fr.nchunk;
			if (ps.cs_off > 0xf0000000ull) {
				return RET_FAILURE_OVERFLOW_ERROR;               // 32-bit offsets: split the batch
			}
		}
		if (ps.luts.empty()) {
			ps.luts.push_back(JhHuffLut());
		}
		lap("luts");
		const size_t cs_bytes = ps.cs_off + 64, seg_entries = ps.seg_off + 4, chunk_entries = ps.chunk_off + 4;
		const size_t sub_entries = ps.sub_off + 4, wg_entries = ps.wg_off + 4;
		HIPCHK(dec->grow((void **)&ps.d_cs, cs_bytes));
		HIPCHK(dec->grow((void **)&ps.d_seg, seg_entries * 4));
		HIPCHK(dec->grow((void **)&ps.d_chunk_kept, chunk_entries * 4));
		HIPCHK(dec->grow((void **)&ps.d_chunk_rst, chunk_entries * 4));
		HIPCHK(dec->grow((void **)&ps.d_start, sub_entries * sizeof(JhSubStart)));
		HIPCHK(dec->grow((void **)&ps.d_rec, sub_entries * (JH_CHAIN_MAX + 1) * sizeof(JhSubInfo)));
		HIPCHK(dec->grow((void **)&ps.d_meta, sub_entries * sizeof(uint32_t)));
		HIPCHK(dec->grow((void **)&ps.d_surv, sub_entries * sizeof(uint32_t)));
		HIPCHK(dec->grow((void **)&ps.d_key01, sub_entries * 2 * sizeof(unsigned long long)));
		HIPCHK(dec->grow((void **)&ps.d_prec, sub_entries * sizeof(JhSubInfo)));
		HIPCHK(dec->grow((void **)&ps.d_wg_key, wg_entries * sizeof(unsigned long long)));
		HIPCHK(dec->grow((void **)&ps.d_wg_agg, wg_entries * sizeof(JhWgAgg)));
		HIPCHK(dec->grow((void **)&ps.d_wg_base, wg_entries * sizeof(JhWgAgg)));
		HIPCHK(hipMemsetAsync(ps.d_meta, 0, sub_entries * sizeof(uint32_t), ups));
		HIPCHK(hipMemsetAsync(ps.d_wg_agg, 0, wg_entries * sizeof(JhWgAgg), ups));
		HIPCHK(hipMemsetAsync(ps.d_wg_base, 0, wg_entries * sizeof(JhWgAgg), ups));
		HIPCHK(dec->grow((void **)&ps.d_frames, nframes * sizeof(JhDecFrame)));
		HIPCHK(dec->grow((void **)&ps.d_state, nframes * sizeof(JhDecState)));
		HIPCHK(dec->grow((void **)&ps.d_luts, ps.luts.size() * sizeof(JhHuffLut)));
		HIPCHK(hipMemsetAsync(ps.d_cs, 0, cs_bytes, ups));
		HIPCHK(hipMemsetAsync(ps.d_seg, 0, seg_entries * 4, ups));
		HIPCHK(dec->grow((void **)&ps.d_drop, seg_entries * sizeof(unsigned long long)));
		HIPCHK(hipMemsetAsync(ps.d_drop, 0, seg_entries * sizeof(unsigned long long), ups));
		// (from pageable memory that stays as it is until the stream has been waited for below)
		HIPCHK(hipMemcpyAsync(ps.d_frames, ps.frames.data(), nframes * sizeof(JhDecFrame), hipMemcpyHostToDevice, ups));
		HIPCHK(hipMemcpyAsync(ps.d_luts, ps.luts.data(), ps.luts.size() * sizeof(JhHuffLut), hipMemcpyHostToDevice, ups));
		if (ps.h_state_cap < (size_t)nframes) {
			if (ps.h_state) {
				hipHostFree(ps.h_state);
				ps.h_state = nullptr;
				ps.h_state_cap = 0;
			}
			const size_t want = (size_t)nframes + (size_t)nframes / 4 + 16;
			HIPCHK(hipHostMalloc((void **)&ps.h_state, want * sizeof(JhDecState), hipHostMallocDefault));
			ps.h_state_cap = want;
		}
		ps.work.rm = rm;
		lap("pass alloc");
	}

	HIPCHK(dec->grow((void **)&dec->d_raw, dec->raw_bytes));
	HIPCHK(dec->grow((void **)&dec->d_quant, nframes * sizeof(JhQuant)));
	HIPCHK(dec->grow((void **)&dec->d_coefs, (size_t)nframes * G.nblk * 64 * sizeof(int16_t)));
	HIPCHK(hipMemcpyAsync(dec->d_quant, quant.data(), nframes * sizeof(JhQuant), hipMemcpyHostToDevice, ups));
	// The tables and zeroed work areas are small; the host arrays they come from may go once this returns.  The
	// stream holds nothing else of this decoder: its last batch has been waited for.
	HIPCHK(hipStreamSynchronize(ups));
	if (direct) {
		// one DMA from the caller's pinned memory, on the decoder's copy stream; run() waits for it on its stream
		const size_t span = (size_t)(span_hi - span_lo);
		if (lead) {
			HIPCHK(hipMemsetAsync(dec->d_raw, 0, lead, ups));
		}
		// in pieces: streams share the process's few hardware queues, and whatever shares one with this stream
		// would wait for a single copy of a hundred megabytes from its first byte to its last
		for (size_t at = 0; at < span; at += JH_UPLOAD_PIECE) {
			const size_t n = span - at < JH_UPLOAD_PIECE ? span - at : (size_t)JH_UPLOAD_PIECE;
			HIPCHK(hipMemcpyAsync(dec->d_raw + lead + at, span_lo + at, n, hipMemcpyHostToDevice, ups));
		}
		HIPCHK(hipMemsetAsync(dec->d_raw + lead + span, 0, dec->raw_bytes - lead - span, ups));
		for (int f = 0; f < nframes; ++f) {
			dec->file_ptr[f] = jpg[f];                         // sync() reads headers there: valid until it has returned
		}
	} else {
		// The codestreams go through ONE pinned staging buffer (kept between calls): the caller's memory is
		// pageable as a rule, and a copy per frame from pageable memory runs at a third of the link's rate.
		// Frames are laid out as on the device, the gaps between them zero.  A few threads fill it piece by piece,
		// and the DMA of a piece starts (copy stream) as soon as the piece and those before it are filled.
		if (dec->stage_bytes < dec->raw_bytes) {
			if (dec->h_stage) {
				hipHostFree(dec->h_stage);
				dec->h_stage = nullptr;
				dec->stage_bytes = 0;
			}
			const size_t want = dec->raw_bytes + dec->raw_bytes / 4;
			HIPCHK(hipHostMalloc((void **)&dec->h_stage, want, hipHostMallocDefault));
			dec->stage_bytes = want;
		}
		uint8_t *stage = dec->h_stage;
		const size_t total = dec->raw_bytes;
		const int nthr = total > (8u << 20) && nframes >= 16 ? fill_threads() : 1;
		const int npiece = nthr == 1 ? 1 : (nframes < 4 * nthr ? nthr : 4 * nthr);
		auto piece_first = [&](int pc) { return (int)((long long)nframes * pc / npiece); };
		std::vector<std::atomic<int>> filled(npiece);
		for (std::atomic<int> &a : filled) {
			a.store(0, std::memory_order_relaxed);
		}
		std::atomic<int> next_piece(0);
		auto fill = [&]() {
			for (;;) {
				const int pc = next_piece.fetch_add(1);
				if (pc >= npiece) {
					return;
				}
				for (int f = piece_first(pc); f < piece_first(pc + 1); ++f) {
					const size_t begin = file_off[f];
					const size_t end = f + 1 < nframes ? file_off[f + 1] : total;
					size_t used = 0;
					if (dec->parse_err[f] == 0) {
						memcpy(stage + begin, jpg[f], jpg_len[f]);
						used = jpg_len[f];
					}
					memset(stage + begin + used, 0, end - begin - used);
				}
				filled[pc].store(1, std::memory_order_release);
			}
		};
		std::vector<std::thread> pool;
		try {
			for (int t = 1; t < nthr; ++t) {
				pool.emplace_back(fill);
			}
		} catch (const std::system_error &) {
			// no more threads to be had: the ones that exist and this one fill every piece (nothing may be thrown
			// through the C boundary)
		}
		hipError_t up = hipSuccess;
		if (pool.empty()) {
			fill();
		}
		for (int pc = 0; pc < npiece; ++pc) {
			while (!filled[pc].load(std::memory_order_acquire)) {
				std::this_thread::yield();
			}
			const size_t a = file_off[piece_first(pc)], b = pc + 1 < npiece ? file_off[piece_first(pc + 1)] : total;
			if (up == hipSuccess && b > a) {
				up = hipMemcpyAsync(dec->d_raw + a, stage + a, b - a, hipMemcpyHostToDevice, ups);
			}
		}
		for (std::thread &th : pool) {
			th.join();
		}
		HIPCHK(up);
		for (int f = 0; f < nframes; ++f) {
			dec->file_ptr[f] = stage + file_off[f];
		}
	}
	HIPCHK(hipEventRecord(dec->uploaded, ups));
	lap("upload");

	for (DecPass &ps : dec->pass) {
		JhDecodeWork &w = ps.work;
		w.raw = dec->d_raw;
		w.cs = ps.d_cs;
		w.seg_start = ps.d_seg;
		w.chunk_kept = ps.d_chunk_kept;
		w.chunk_rst = ps.d_chunk_rst;
		w.rec = ps.d_rec;
		w.meta = ps.d_meta;
		w.surv = ps.d_surv;
		w.key01 = ps.d_key01;
		w.prec = ps.d_prec;
		w.wg_key = ps.d_wg_key;
		w.wg_agg = ps.d_wg_agg;
		w.wg_base = ps.d_wg_base;
		w.start = ps.d_start;
		w.drop = ps.d_drop;
		w.frames = ps.d_frames;
		w.state = ps.d_state;
		w.luts = ps.d_luts;
		w.coefs = dec->d_coefs;
		w.nframes = nframes;
		w.max_chunks = ps.max_chunks;
		w.max_subs = ps.max_subs;
		w.max_luts = ps.max_luts;
		w.lut_bytes = ps.lut_bytes;
		w.debug = dec->debug;
		w.segmented = ps.segmented;
		w.sub_shift = ps.sub_shift;
	}
	dec->nframes = nframes;
	return RET_SUCCESS;
}

// all_scans: every scan is found on the host, also where the first one holds every component (see parse_frame)
static int prepare_staged(jpeg_hip_decoder *dec, int nframes, const uint8_t *const *jpg, const size_t *jpg_len, bool all_scans)
{
	std::vector<const uint8_t *> file(jpg, jpg + nframes);
	std::vector<size_t> file_len(jpg_len, jpg_len + nframes);
	std::vector<std::vector<uint8_t>> patched(nframes);    // damaged files with several scans, see parse_frame()
	return decoder_prepare(dec, nframes, file.data(), file_len.data(), [&](int f, JhHeader *h, std::vector<ScanDesc> *sc) {
		int err = RET_FAILURE_FILE_UNSUPPORTED;
		for (int tries = 0; tries < 64; ++tries) {
			size_t cut[2] = {0, 0};
			err = parse_frame(file[f], file_len[f], h, sc, dec->tolerant, cut, all_scans);
			if (err != PARSE_SPLICE) {
				return err;
			}
			std::vector<uint8_t> &v = patched[f];
			if (v.empty()) {
				v.assign(file[f], file[f] + file_len[f]);
			}
			v.erase(v.begin() + cut[0], v.begin() + cut[1]);
			file[f] = v.data();
			file_len[f] = v.size();
		}
		return err == PARSE_SPLICE ? RET_FAILURE_FILE_UNSUPPORTED : err;
	});
}

extern "C" int jpeg_hip_decoder_prepare(jpeg_hip_decoder *dec, int nframes,
                                        const uint8_t *const *jpg, const size_t *jpg_len)
{
	if (!dec || nframes <= 0 || !jpg || !jpg_len) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	dec->all_scans_tried = false;
	return prepare_staged(dec, nframes, jpg, jpg_len, false);
}

extern "C" int jpeg_hip_decoder_prepare_pinned(jpeg_hip_decoder *dec, int nframes,
                                               const uint8_t *const *jpg, const size_t *jpg_len)
{
	if (!dec || nframes <= 0 || !jpg || !jpg_len) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	dec->all_scans_tried = false;
	bool pinned = true, patch = false;
	int err = decoder_prepare(dec, nframes, jpg, jpg_len, [&](int f, JhHeader *h, std::vector<ScanDesc> *sc) {
		size_t cut[2] = {0, 0};
		const int e = parse_frame(jpg[f], jpg_len[f], h, sc, dec->tolerant, cut);
		if (e == PARSE_SPLICE) {
			patch = true;                                  // a damaged frame that needs a patched copy: the staged path makes it
			pinned = false;
		}
		return e;
	}, &pinned);
	if (patch) {
		err = jpeg_hip_decoder_prepare(dec, nframes, jpg, jpg_len);
	}
	return err;
}

extern "C" void *jpeg_hip_host_alloc(size_t bytes)
{
	void *p = nullptr;
	return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

extern "C" void jpeg_hip_host_free(void *p)
{
	if (p) {
		hipHostFree(p);
	}
}

// ======================================================================================= streams on different hardware queues
//
// A process has four hardware queues (GPU_MAX_HW_QUEUES) and HIP maps every stream to one of them as streams come
// into being: A B C D D C B A A B ... in order (measured, tools/queue_probe.hip).  Work on two streams of one queue
// runs one piece after the other, so a pipeline that wants its batches side by side must not take "the next three
// streams": which queues those land on depends on how many streams the process has made before (three in a row
// from position 3 are D D C -- the Motion JPEG pipeline then ran at 46 k instead of 72 k frames/s).  This makes
// candidates and MEASURES: two kernels that note the time when they start and end, one per stream; streams whose
// kernels ran side by side are on different queues.
__global__ void k_queue_probe(unsigned long long *stamp, long long ticks)
{
	const long long t0 = wall_clock64();
	if (threadIdx.x == 0) {
		stamp[0] = (unsigned long long)t0;
	}
	while (wall_clock64() - t0 < ticks) {
	}
	if (threadIdx.x == 0) {
		stamp[1] = (unsigned long long)wall_clock64();
	}
}

// did a kernel on b start before the kernel on a, launched first, ended?
static bool streams_overlap(hipStream_t a, hipStream_t b, unsigned long long *h_stamp)
{
	const long long ticks = 20000;                         // wall_clock64() counts at 100 MHz: 200 us
	memset(h_stamp, 0, 4 * sizeof *h_stamp);
	hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, a, h_stamp, ticks);
	hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, b, h_stamp + 2, ticks);
	if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) {
		return false;
	}
	return h_stamp[2] < h_stamp[1] && h_stamp[0] < h_stamp[3];
}

extern "C" int jpeg_hip_streams_create(int device, int n, void *avoid, int use_avoid, void **streams)
{
	int err = require_device(device);
	if (err) {
		return -1;
	}
	if (n < 1 || n > 4 || !streams) {
		return -1;
	}
	const int ncand = 8;                                   // one period of the mapping: every queue is among them
	hipStream_t cand[ncand];
	int made = 0;
	for (; made < ncand; ++made) {
		if (hipStreamCreateWithFlags(&cand[made], hipStreamNonBlocking) != hipSuccess) {
			break;
		}
	}
	unsigned long long *h_stamp = nullptr;
	if (made < n || hipHostMalloc((void **)&h_stamp, 4 * sizeof *h_stamp, hipHostMallocDefault) != hipSuccess) {
		for (int i = 0; i < made; ++i) {
			hipStreamDestroy(cand[i]);
		}
		(void)hipGetLastError();
		return -1;
	}
	bool keep[ncand] = {false};
	int kept[4], nkept = 0;
	for (int c = 0; c < made && nkept < n; ++c) {
		bool ok = !use_avoid || streams_overlap((hipStream_t)avoid, cand[c], h_stamp);
		for (int k = 0; k < nkept && ok; ++k) {
			ok = streams_overlap(cand[kept[k]], cand[c], h_stamp);
		}
		if (ok) {
			kept[nkept++] = c;
			keep[c] = true;
		}
	}
	const int proven = nkept;
	for (int c = 0; c < made && nkept < n; ++c) {          // fewer free queues than asked for: the rest share
		if (!keep[c]) {
			kept[nkept++] = c;
			keep[c] = true;
		}
	}
	for (int c = 0; c < made; ++c) {
		if (!keep[c]) {
			hipStreamDestroy(cand[c]);
		}
	}
	hipHostFree(h_stamp);
	for (int k = 0; k < n; ++k) {
		streams[k] = (void *)cand[kept[k]];
	}
	return proven;
}

extern "C" void jpeg_hip_streams_destroy(int device, int n, void **streams)
{
	if (!streams || hipSetDevice(device) != hipSuccess) {
		return;
	}
	for (int k = 0; k < n; ++k) {
		if (streams[k]) {
			hipStreamSynchronize((hipStream_t)streams[k]);
			hipStreamDestroy((hipStream_t)streams[k]);
			streams[k] = nullptr;
		}
	}
}

extern "C" int jpeg_hip_decoder_info(const jpeg_hip_decoder *dec, jpeg_hip_frame_info *info)
{
	if (!dec || dec->nframes == 0) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	fill_info(dec->g, info);
	return RET_SUCCESS;
}

extern "C" size_t jpeg_hip_decoder_frame_bytes(const jpeg_hip_decoder *dec)
{
	const JhGeom &g = dec->g;
	return (size_t)g.X * g.Y * g.out_comps * g.sample_bytes;
}

__global__ void k_dec_init(JhDecState *st, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		JhDecState s;
		memset(&s, 0, sizeof s);
		s.term_pos = 0xffffffffu;
		st[i] = s;
	}
}

// frames [f0, f1) of the batch on stream s: every pass, then -- behind `before_pixels`, if given -- the pixel
// kernel, and the states' way back
static int decoder_launch_range(jpeg_hip_decoder *dec, int f0, int f1, hipStream_t s, const JhTimerHook *th, hipEvent_t before_pixels)
{
	const int n = f1 - f0;
	if (dec->pass.size() != 1 || dec->pass[0].comp != -1) {
		// scans of single components: blocks no scan reaches stay zero (common.c:129)
		if (hipMemsetAsync(dec->d_coefs + (size_t)f0 * dec->g.nblk * 64, 0, (size_t)n * dec->g.nblk * 64 * sizeof(int16_t), s) != hipSuccess) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	for (DecPass &ps : dec->pass) {
		JhDecodeWork w = ps.work;                          // the kernels index frames[], state[] and coefs by frame
		w.frames += f0;
		w.state += f0;
		w.coefs += (size_t)f0 * (w.rm.mode ? w.rm.nblk_full : ps.g.nblk) * 64;
		w.nframes = n;
		hipLaunchKernelGGL(k_dec_init, dim3((n + 63) / 64), dim3(64), 0, s, ps.d_state + f0, n);
		if (jh_launch_ecs_unstuff(w, s, th)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		if (jh_launch_huff_decode(w, ps.g, s, th)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	if (before_pixels && hipStreamWaitEvent(s, before_pixels, 0) != hipSuccess) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (!dec->coefs_only) {                                // jpeg_hip_read_scan: the coefficients are the result
		dec->timer.begin("k_decode_pixels", s);
		if (jh_launch_decode_pixels(dec->g, n, dec->d_coefs + (size_t)f0 * dec->g.nblk * 64, dec->d_quant + f0,
		                            (uint8_t *)dec->last_out + (size_t)f0 * dec->last_stride, dec->last_stride, s, dec->float_tap)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		dec->timer.end(s);
	}
	// the frames' states come back on the same stream, into page-locked memory: sync() needs the event only
	for (DecPass &ps : dec->pass) {
		if (hipMemcpyAsync(ps.h_state + f0, ps.d_state + f0, n * sizeof(JhDecState), hipMemcpyDeviceToHost, s) != hipSuccess) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	return RET_SUCCESS;
}

// run(): the batch on the caller's stream s, or laid out over the lanes.  A lane other than the first is a stream
// of its own that works on the same frames in every run().  Its entropy kernels touch nothing but those frames'
// part of the decoder's buffers, so they wait for the codestreams' upload only and may start while s -- and the
// other lanes -- are still busy with the run() before: back-to-back calls then drift apart the way separate
// decoder objects would, narrow kernels beside wide ones.  What the caller's stream orders -- the output buffer,
// whatever was queued on s before this call -- is waited for in front of the pixel kernel, and s waits for the
// lanes' ends.  (Upload on the caller's stream, set_upload_stream: the order of s is all there is; the lanes wait
// for it at their start.)
static int decoder_launch(jpeg_hip_decoder *dec, hipStream_t s)
{
	const JhTimerHook th = dec->timer.hook();
	// lanes: not while kernels are timed one by one or a stage test taps the float frame (both count on one stream)
	const int lanes = dec->timer.on || dec->float_tap ? 1 : std::min(dec->lanes, dec->nframes);
	if (lanes > 1 && hipEventRecord(dec->lane_fork, s) != hipSuccess) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	for (int k = 0; k < lanes; ++k) {
		const int f0 = (int)((long long)dec->nframes * k / lanes), f1 = (int)((long long)dec->nframes * (k + 1) / lanes);
		const hipStream_t ls = k == 0 ? s : dec->lane_stream[k - 1];
		hipEvent_t before_pixels = nullptr;
		if (k > 0) {
			if (dec->up_user) {
				if (hipStreamWaitEvent(ls, dec->lane_fork, 0) != hipSuccess) {
					return RET_FAILURE_LOGIC_ERROR;
				}
			} else {
				if (hipStreamWaitEvent(ls, dec->uploaded, 0) != hipSuccess) {
					return RET_FAILURE_LOGIC_ERROR;
				}
				before_pixels = dec->lane_fork;
			}
		}
		const int err = decoder_launch_range(dec, f0, f1, ls, &th, before_pixels);
		if (err) {
			return err;
		}
		if (k > 0 && hipEventRecord(dec->lane_done[k - 1], ls) != hipSuccess) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	for (int k = 1; k < lanes; ++k) {
		if (hipStreamWaitEvent(s, dec->lane_done[k - 1], 0) != hipSuccess) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	return hipEventRecord(dec->done, s) == hipSuccess ? RET_SUCCESS : RET_FAILURE_LOGIC_ERROR;
}

// Lanes.  n = 1 ... 4: run() splits the batch into n runs of frames, the first on its own stream, the others on
// streams of this object that are measured to run beside `stream` and beside each other (jpeg_hip_streams_create;
// `stream` = the stream run() will be called with).  Results, order on the caller's stream and sync() are as
// before: the lanes start behind what the caller's stream holds when run() is called, and the stream waits for
// them.  lane_streams: NULL, or n - 1 streams of the caller's to use instead (a pipeline that lays out the
// process's four hardware queues itself, with jpeg_hip_streams_create: bench.py; they must outlive their use here).
// Returns the number of lanes that were seen to run side by side (n, or fewer when the process's hardware queues
// are taken; n unmeasured for streams handed in), negative on error.
extern "C" int jpeg_hip_decoder_set_lanes(jpeg_hip_decoder *dec, int n, void *stream, void *const *lane_streams)
{
	if (!dec || n < 1 || n > JH_MAX_LANES) {
		return -1;
	}
	if (hipSetDevice(dec->device) != hipSuccess) {
		return -1;
	}
	if (dec->last_stream) {
		hipStreamSynchronize(dec->last_stream);            // a run() in flight may be using the lanes
	}
	dec->lanes_chosen = true;
	decoder_drop_lanes(dec);
	if (n == 1) {
		return 1;
	}
	void *made[JH_MAX_LANES - 1] = {nullptr, nullptr, nullptr};
	int proven = n - 1;
	dec->lanes_owned = lane_streams == nullptr;
	if (lane_streams) {
		for (int k = 0; k < n - 1; ++k) {
			made[k] = lane_streams[k];
		}
	} else {
		proven = jpeg_hip_streams_create(dec->device, n - 1, stream, 1, made);
		if (proven < 0) {
			return -1;
		}
	}
	bool ok = hipEventCreateWithFlags(&dec->lane_fork, hipEventDisableTiming) == hipSuccess;
	for (int k = 0; k < n - 1; ++k) {
		dec->lane_stream[k] = (hipStream_t)made[k];
		ok = ok && hipEventCreateWithFlags(&dec->lane_done[k], hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) {
		decoder_drop_lanes(dec);
		return -1;
	}
	dec->lanes = n;
	return 1 + proven;
}

extern "C" int jpeg_hip_decoder_run(jpeg_hip_decoder *dec, void *d_samples, size_t frame_stride, void *stream)
{
	if (!dec || dec->nframes == 0 || !d_samples) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (frame_stride < jpeg_hip_decoder_frame_bytes(dec)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(dec->device));
	dec->last_out = d_samples;
	dec->last_stride = frame_stride;
	if (!dec->lanes_chosen) {
		dec->lanes_chosen = true;
		const char *e = getenv("JPEG_AMD_LANES");
		if (e && atoi(e) > 1 && jpeg_hip_decoder_set_lanes(dec, std::min(atoi(e), JH_MAX_LANES), stream, nullptr) < 0) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	dec->last_stream = (hipStream_t)stream;
	dec->timer.reset();
	HIPCHK(hipStreamWaitEvent(dec->last_stream, dec->uploaded, 0));    // the codestreams' DMA (copy stream)
	int err = decoder_launch(dec, dec->last_stream);
	dec->ran = err == RET_SUCCESS;
	return err;
}

extern "C" int jpeg_hip_decoder_sync(jpeg_hip_decoder *dec, int *frame_status)
{
	if (!dec || !dec->ran) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	for (int f = 0; frame_status && f < dec->nframes; ++f) {
		frame_status[f] = RET_FAILURE_LOGIC_ERROR;         // what every early return below leaves behind
	}
	HIPCHK(hipSetDevice(dec->device));
	HIPCHK(hipEventSynchronize(dec->done));                // this run's kernels, not what the caller queued behind them
	for (int attempt = 0;; ++attempt) {
		for (DecPass &ps : dec->pass) {
			memcpy(ps.state.data(), ps.h_state, dec->nframes * sizeof(JhDecState));   // run() brought them over
		}
		// The restart-segment tables are sized from DRI.  The reference never looks at Ri (quirk Q1): the
		// markers decide, and a file may hold more of them than its header promised (or none was sent).
		// The device counted them all the same: size the tables from that count and run once more.
		bool again = false;
		for (DecPass &ps : dec->pass) {
			bool grown = false;
			for (int f = 0; f < dec->nframes; ++f) {
				if (!dec->parse_err[f] && (ps.state[f].status & JH_ST_SEG_OVERFLOW)) {
					ps.frames[f].seg_cap = ps.state[f].nseg + 64;
					grown = true;
				}
			}
			if (!grown) {
				continue;
			}
			again = true;
			size_t off = 0;
			for (int f = 0; f < dec->nframes; ++f) {
				ps.frames[f].seg_off = (uint32_t)off;
				off += ps.frames[f].seg_cap;
			}
			ps.seg_off = off;
			const size_t seg_entries = off + 4;
			HIPCHK(dec->grow((void **)&ps.d_seg, seg_entries * 4));
			HIPCHK(dec->grow((void **)&ps.d_drop, seg_entries * sizeof(unsigned long long)));
			HIPCHK(hipMemset(ps.d_seg, 0, seg_entries * 4));
			HIPCHK(hipMemset(ps.d_drop, 0, seg_entries * sizeof(unsigned long long)));
			HIPCHK(hipMemcpy(ps.d_frames, ps.frames.data(), dec->nframes * sizeof(JhDecFrame), hipMemcpyHostToDevice));
			ps.work.seg_start = ps.d_seg;
			ps.work.drop = ps.d_drop;
		}
		// A marker inside the data of a scan that is neither RSTn nor EOI ends read_ecs() (io.c:257-268);
		// parse_format() goes on from it (decoder.c:487-657), skips what it can skip (TEM, APPn, COM, DAC, DRI
		// segments by their length, bytes between markers) and, when it arrives at an RSTn, decodes on from
		// there (decoder.c:622-633) with the MCU counter where it stood.  For the decoder that is the stream with
		// the skipped bytes cut out: done here on the host copy of the frame, and the batch runs once more.
		if (!again && dec->pass.size() == 1 && dec->pass[0].comp == -1 && !dec->coefs_only) {
			DecPass &ps = dec->pass[0];
			bool cut = false;
			for (int f = 0; f < dec->nframes; ++f) {
				const JhDecState &ds = ps.state[f];
				JhDecFrame &fr = ps.frames[f];
				if (dec->parse_err[f] || (ds.status & JH_ST_SEG_OVERFLOW) || ds.term_pos + 1 >= fr.raw_end || ds.term_marker == 0xd9) {
					continue;
				}
				const uint8_t *orig = dec->file_ptr[f];
				const size_t len = dec->file_len[f], rel = ds.term_pos - dec->file_off[f];
				size_t rst = 0;
				if (!orig || rel >= len || jh_walk_to_restart(orig, len, rel, dec->tolerant, &rst) != RET_SUCCESS || rst <= rel) {
					continue;                                  // EOI, or something that ends the decoding: settled below
				}
				const size_t removed = rst - rel;
				uint8_t *file;
				if (dec->h_stage && orig >= dec->h_stage && orig < dec->h_stage + dec->stage_bytes) {
					file = dec->h_stage + (orig - dec->h_stage);       // our staging copy
				} else {
					if (dec->patched[f].empty()) {
						dec->patched[f].assign(orig, orig + len);      // the caller's memory is not ours to change
					}
					file = dec->patched[f].data();
					dec->file_ptr[f] = file;
				}
				memmove(file + rel, file + rst, len - rst);
				memset(file + len - removed, 0, removed);
				HIPCHK(hipMemcpy(dec->d_raw + dec->file_off[f] + rel, file + rel, len - rel, hipMemcpyHostToDevice));
				dec->file_len[f] = len - removed;
				dec->spliced[f] += removed;
				fr.raw_end -= (uint32_t)removed;
				cut = true;
			}
			if (cut) {
				HIPCHK(hipMemcpy(ps.d_frames, ps.frames.data(), dec->nframes * sizeof(JhDecFrame), hipMemcpyHostToDevice));
				again = true;
			}
		}
		if (!again || attempt >= 64) {
			break;
		}
		const int err = decoder_launch(dec, dec->last_stream);
		if (err) {
			return err;
		}
		HIPCHK(hipEventSynchronize(dec->done));
	}
	int worst = RET_SUCCESS;
	bool more_scans = false;
	for (int f = 0; f < dec->nframes; ++f) {
		int st = dec->parse_err[f];
		dec->ecs_len[f] = 0;
		for (size_t pi = 0; pi < dec->pass.size() && !st; ++pi) {
			const JhDecState &ds = dec->pass[pi].state[f];
			const JhDecFrame &fr = dec->pass[pi].frames[f];
			dec->ecs_len[f] += ds.term_pos - fr.raw_begin + dec->spliced[f];
			if (ds.status & JH_ST_SEG_OVERFLOW) {
				st = RET_FAILURE_FILE_UNSUPPORTED;
			} else if ((ds.status & JH_ST_COEF_RANGE) && dec->g.P > 12) {
				st = RET_FAILURE_FILE_UNSUPPORTED;   // a DC coefficient beyond 16 bits (P 13 ... 16 only; jpeg_amd.h, limits)
			} else if (ds.term_pos + 1 >= fr.raw_end) {
				st = RET_FAILURE_FILE_IO;            // no EOI (or a lone FF as the last byte): read_marker() hits end of file (io.c:197)
			} else if (pi + 1 == dec->pass.size() && dec->pass[pi].comp == -1 && ds.term_marker != 0xd9 && !dec->coefs_only) {
				// A scan of all components that ends on something else than EOI.  parse_format() goes on from
				// that marker (decoder.c:487-657): TEM, APPn, COM, DAC and table segments pass, whatever
				// lies between markers is skipped (io.c:196-210), EOI ends the image with what was decoded.
				// The same walk here, on the host copy of the file (restart intervals behind such a marker have
				// been spliced in above).  Another SOS behind it sends the batch through prepare once more (below);
				// new quantisation tables would change the image already made and are refused.
				st = RET_FAILURE_FILE_UNSUPPORTED;
				JhHeader h0, h1;
				const uint8_t *file = dec->file_ptr[f];
				const size_t len = dec->file_len[f];
				bool eoi = false;
				if (file && jh_parse_headers(file, len, &h0, dec->tolerant) == RET_SUCCESS) {
					h1 = h0;
					const int e = jh_parse_next_scan(file, len, ds.term_pos - dec->file_off[f], &h1, &eoi);
					if (e == RET_SUCCESS && eoi && memcmp(h0.Q, h1.Q, sizeof h0.Q) == 0) {
						st = RET_SUCCESS;
					} else if (e == RET_SUCCESS && !eoi) {
						more_scans = true;               // another SOS: the components are coded again (decoder.c:593-601)
					} else if (e == RET_FAILURE_FILE_IO) {
						st = e;                          // no EOI behind it: read_marker() meets the end of the file
					}
				}
			}
		}
		if (frame_status) {
			frame_status[f] = st;
		}
		if (st && !worst) {
			worst = st;
		}
	}
	if (more_scans && !dec->all_scans_tried && !dec->coefs_only) {
		// A frame whose first scan holds every component goes on with another scan.  prepare() left the end of that
		// first scan to the device; now the host walks the files scan by scan and the batch is decoded again, one
		// pass per scan (frames of the batch that differ in their scans are refused, as always).
		dec->all_scans_tried = true;
		const int n = dec->nframes;
		std::vector<std::vector<uint8_t>> copy(n);
		std::vector<const uint8_t *> ptr(n);
		std::vector<size_t> len(n);
		for (int f = 0; f < n; ++f) {
			if (dec->file_ptr[f]) {
				copy[f].assign(dec->file_ptr[f], dec->file_ptr[f] + dec->file_len[f]);   // the staging buffer is about to be refilled
			}
			ptr[f] = copy[f].data();
			len[f] = copy[f].size();
		}
		void *out = dec->last_out;
		const size_t stride = dec->last_stride;
		hipStream_t s = dec->last_stream;
		int err = prepare_staged(dec, n, ptr.data(), len.data(), true);
		if (!err) {
			err = jpeg_hip_decoder_run(dec, out, stride, s);
		}
		if (err) {
			for (int f = 0; frame_status && f < n; ++f) {
				frame_status[f] = err;
			}
			return err;
		}
		return jpeg_hip_decoder_sync(dec, frame_status);
	}
	return worst;
}

extern "C" int jpeg_hip_decoder_get_coefficients(jpeg_hip_decoder *dec, int frame, int k, int32_t *out)
{
	if (!dec || !dec->ran || frame < 0 || frame >= dec->nframes || k < 0 || k >= dec->g.ncomp) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(dec->device));
	const size_t n = (size_t)dec->g.nblk * 64;
	std::vector<int16_t> fc(n);
	HIPCHK(hipMemcpy(fc.data(), dec->d_coefs + (size_t)frame * n, n * sizeof(int16_t), hipMemcpyDeviceToHost));
	coefs_to_reference_layout(dec->g, fc.data(), k, out);
	return RET_SUCCESS;
}

extern "C" void jpeg_hip_decoder_set_timing(jpeg_hip_decoder *dec, int on) { dec->timer.on = on != 0; }

extern "C" int jpeg_hip_decoder_get_timing(jpeg_hip_decoder *dec, const char **names, float *ms, int cap)
{
	hipSetDevice(dec->device);
	return dec->timer.collect(names, ms, cap);
}

extern "C" int jpeg_hip_decoder_get_stats(const jpeg_hip_decoder *dec, int frame, uint32_t stats[4])
{
	if (!dec || !dec->ran || frame < 0 || frame >= dec->nframes) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	stats[0] = stats[1] = stats[2] = stats[3] = 0;
	for (const DecPass &ps : dec->pass) {                  // summed over the scans of the frame
		const JhDecState &s = ps.state[frame];
		stats[0] |= (s.status & JH_ST_SERIAL) ? (s.serial_cause ? s.serial_cause : 1u) : 0u;
		stats[1] += s.nrepaired;
		stats[2] += s.nblocks;
		stats[3] += s.nseg;
	}
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_decoder_get_chain_steps(jpeg_hip_decoder *dec, int frame, uint32_t *hist, int cap)
{
	if (!dec || !dec->ran || frame < 0 || frame >= dec->nframes || !hist || cap <= 0 || dec->pass.empty()) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(dec->device));
	const DecPass &ps = dec->pass[0];
	const uint32_t nsub = (ps.state[frame].cs_len * 8u + (1u << ps.sub_shift) - 1u) >> ps.sub_shift;
	std::vector<uint32_t> meta(nsub);
	if (nsub) {
		HIPCHK(hipMemcpy(meta.data(), ps.d_meta + ps.frames[frame].sub_off, nsub * sizeof(uint32_t), hipMemcpyDeviceToHost));
	}
	for (int i = 0; i < cap; ++i) {
		hist[i] = 0;
	}
	for (uint32_t m : meta) {
		const int steps = (int)(m & 0xffu);
		hist[steps < cap ? steps : cap - 1]++;
	}
	return RET_SUCCESS;
}

extern "C" void jpeg_hip_decoder_set_float_tap(jpeg_hip_decoder *dec, void *d_floats)
{
	if (dec) {
		dec->float_tap = (float *)d_floats;
	}
}

extern "C" void jpeg_hip_decoder_set_debug(jpeg_hip_decoder *dec, int flags)
{
	dec->debug = flags;
	for (DecPass &ps : dec->pass) {
		ps.work.debug = flags;
	}
}

extern "C" void jpeg_hip_decoder_set_upload_stream(jpeg_hip_decoder *dec, void *stream, int on)
{
	dec->up_user = on != 0;
	dec->up_stream = on ? (hipStream_t)stream : nullptr;
}

extern "C" void jpeg_hip_decoder_set_tolerant(jpeg_hip_decoder *dec, int on)
{
	dec->tolerant = on != 0;
}

extern "C" int jpeg_hip_decoder_ecs_bytes(const jpeg_hip_decoder *dec, size_t *ecs_len)
{
	for (int f = 0; f < dec->nframes; ++f) {
		ecs_len[f] = dec->ecs_len[f];
	}
	return RET_SUCCESS;
}

// ======================================================================================= encoder

struct jpeg_hip_encoder {
	int device = 0;
	int nframes = 0;
	JhGeom g;
	jpeg_hip_params prm;
	int restart_interval = 0;         // EXTENSION, set by jpeg_hip_encoder_set_restart_interval(); read by configure()
	uint16_t Q[4][64];
	JhHuffSpec ht[2][4];
	std::vector<std::vector<uint8_t>> headers;   // per frame
	std::vector<uint32_t> hdr_len;
	std::vector<JhEncState> state;
	// device
	JhQuant *d_quant = nullptr;
	int16_t *d_coefs = nullptr;
	JhHuffEnc *d_codes = nullptr;
	JhRunTab *d_runtab = nullptr;         // [frame]: the pixel kernel's price tables (default tables only)
	JhDevSpec *d_specs = nullptr;
	uint8_t *d_hdr_parts = nullptr;   // header bytes around the DHT segments (optimised mode builds headers on the device)
	uint32_t *d_hist = nullptr, *d_wg_bits = nullptr, *d_us = nullptr, *d_chunk_ff = nullptr, *d_hdr_len = nullptr;
	uint32_t *d_ivl_raw = nullptr, *d_ivl_padsum = nullptr, *d_ivl_eb = nullptr;
	uint16_t *d_bitlen = nullptr;
	unsigned long long *d_nzmap = nullptr;
	uint32_t *d_acdc = nullptr;
	uint8_t *d_out = nullptr;
	JhEncState *d_state = nullptr;
	unsigned long long *d_soff = nullptr;   // stream out: where each frame went ([nframes + 1]) and how long it is
	uint32_t *d_slen = nullptr;
	// page-locked copies of what sync() reads: run() ends with asynchronous copies into them and an event
	JhEncState *h_state = nullptr;
	uint32_t *h_hdr_len = nullptr, *h_slen = nullptr;
	unsigned long long *h_soff = nullptr;
	hipEvent_t done = nullptr;
	void *stream_dst = nullptr;       // run_stream(): where the kernels pack the codestreams (device memory), NULL after a plain run()
	size_t stream_cap = 0;
	float *float_tap = nullptr;       // stage tests: see jpeg_hip_encoder_set_float_tap()
	void *host_dst = nullptr;         // run_stream() into page-locked host memory: sync_stream() sends the packed stream there
	uint8_t *d_pack = nullptr;        // ... from this staging buffer, by DMA on copy_stream
	size_t d_pack_cap = 0;
	hipStream_t copy_stream = nullptr;
	hipEvent_t copied = nullptr;
	JhEncodeWork work;
	hipStream_t last_stream = nullptr;
	bool ran = false;
	long long alloc_sig[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // what the device buffers were sized for
	bool acbits_ready = false;                           // the last run's pixel kernel filled d_nzmap / d_acdc (default tables)
	size_t grown_us = 0, grown_out = 0;                  // stream strides after sync() had to grow them (kept while alloc_sig holds)
	StageTimer timer;

	void release()
	{
		hipFree(d_quant); hipFree(d_coefs); hipFree(d_codes); hipFree(d_runtab); hipFree(d_hist); hipFree(d_wg_bits);
		hipFree(d_us); hipFree(d_chunk_ff); hipFree(d_hdr_len); hipFree(d_bitlen); hipFree(d_nzmap); hipFree(d_acdc); hipFree(d_out); hipFree(d_state);
		hipFree(d_ivl_raw); hipFree(d_ivl_padsum); hipFree(d_ivl_eb); hipFree(d_specs); hipFree(d_hdr_parts);
		hipFree(d_soff); hipFree(d_slen); hipFree(d_pack);
		d_pack = nullptr; d_pack_cap = 0;
		if (h_state) { hipHostFree(h_state); }
		if (h_hdr_len) { hipHostFree(h_hdr_len); }
		if (h_slen) { hipHostFree(h_slen); }
		if (h_soff) { hipHostFree(h_soff); }
		d_soff = nullptr; d_slen = nullptr; h_state = nullptr; h_hdr_len = h_slen = nullptr; h_soff = nullptr;
		d_specs = nullptr; d_hdr_parts = nullptr;
		d_ivl_raw = d_ivl_padsum = d_ivl_eb = nullptr;
		d_quant = nullptr; d_coefs = nullptr; d_codes = nullptr; d_runtab = nullptr; d_hist = d_wg_bits = d_us = d_chunk_ff = d_hdr_len = nullptr;
		d_bitlen = nullptr; d_nzmap = nullptr; d_acdc = nullptr; d_out = nullptr; d_state = nullptr;
	}
};

extern "C" int jpeg_hip_encoder_create(jpeg_hip_encoder **enc, int device)
{
	*enc = nullptr;
	int err = require_device(device);
	if (err) {
		return err;
	}
	jpeg_hip_encoder *e = new jpeg_hip_encoder();
	e->device = device;
	memset(&e->g, 0, sizeof e->g);
	if (hipSetDevice(device) != hipSuccess || hipEventCreateWithFlags(&e->done, hipEventDisableTiming) != hipSuccess) {
		delete e;
		return RET_FAILURE_LOGIC_ERROR;
	}
	*enc = e;
	return RET_SUCCESS;
}

extern "C" void jpeg_hip_encoder_destroy(jpeg_hip_encoder *enc)
{
	if (enc) {
		hipSetDevice(enc->device);
		if (enc->ran) {
			hipEventSynchronize(enc->done);
		}
		enc->release();
		if (enc->done) {
			hipEventDestroy(enc->done);
		}
		if (enc->copied) {
			hipEventDestroy(enc->copied);
		}
		if (enc->copy_stream) {
			hipStreamDestroy(enc->copy_stream);
		}
		delete enc;
	}
}

// default tables (e->ht) and the headers that carry them, for every frame; the optimised mode replaces both
// on the device in every run (k_enc_adapt, k_enc_headers)
static int upload_codes_and_headers(jpeg_hip_encoder *e, int f0, int f1, hipStream_t s)
{
	std::vector<JhHuffEnc> codes((size_t)(f1 - f0) * 4);
	for (int f = f0; f < f1; ++f) {
		const JhHuffSpec(&ht)[2][4] = e->ht;
		for (int tc = 0; tc < 2; ++tc) {
			for (int th = 0; th < 2; ++th) {
				int err = jh_build_enccodes(&ht[tc][th], &codes[(size_t)(f - f0) * 4 + tc * 2 + th]);
				if (err) {
					return err;
				}
			}
		}
		jh_produce_headers(e->g, e->Q, ht, e->restart_interval, &e->headers[f]);
		e->hdr_len[f] = (uint32_t)e->headers[f].size();
		if (e->hdr_len[f] > 2048) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		HIPCHK(hipMemcpyAsync(e->d_out + (size_t)f * e->work.out_stride, e->headers[f].data(), e->hdr_len[f],
		                      hipMemcpyHostToDevice, s));
	}
	HIPCHK(hipMemcpyAsync(e->d_codes + (size_t)f0 * 4, codes.data(), codes.size() * sizeof(JhHuffEnc), hipMemcpyHostToDevice, s));
	HIPCHK(hipMemcpyAsync(e->d_hdr_len + f0, e->hdr_len.data() + f0, (size_t)(f1 - f0) * 4, hipMemcpyHostToDevice, s));
	HIPCHK(hipStreamSynchronize(s));   // host vectors above are transient
	return RET_SUCCESS;
}

// the header bytes in front of the DHT segments [0, pre) and behind them [suf, size) do not depend on the tables
static bool split_headers(const std::vector<uint8_t> &full, size_t *pre_out, size_t *suf_out)
{
	size_t pos = 2, pre = 0, suf = 0;
	while (pos + 4 <= full.size()) {
		const unsigned m = full[pos + 1];
		if (m == 0xc4 && pre == 0) {
			pre = pos;
		}
		if (pre != 0 && m != 0xc4) {
			suf = pos;
			break;
		}
		pos += 2 + (((size_t)full[pos + 2] << 8) | full[pos + 3]);
	}
	*pre_out = pre;
	*suf_out = suf;
	return pre != 0 && suf != 0 && full.size() <= 4096;
}

// read_image() component set-up (encoder.c:109-146): ids 1..3, luma H x V, chroma 1 x 1
static int encoder_geometry(int X, int Y, int Nf, int P, const jpeg_hip_params *params, JhGeom *g)
{
	if ((Nf != 1 && Nf != 3) || X <= 0 || Y <= 0 || X > 65535 || Y > 65535) {
		return RET_FAILURE_FILE_UNSUPPORTED;               // encoder.c:147-148
	}
	if (Nf == 3 && (params->H < 1 || params->H > 2 || params->V < 1 || params->V > 2)) {
		return RET_FAILURE_FILE_UNSUPPORTED;               // assert in the reference (encoder.c:123-124)
	}
	memset(g, 0, sizeof *g);
	g->P = P; g->X = X; g->Y = Y; g->Nf = Nf; g->ncomp = Nf;
	for (int k = 0; k < Nf; ++k) {
		g->id[k] = k + 1;
		g->H[k] = (k == 0 && Nf == 3) ? params->H : 1;
		g->V[k] = (k == 0 && Nf == 3) ? params->V : 1;
		g->Tq[k] = g->Td[k] = g->Ta[k] = k == 0 ? 0 : 1;
		g->scan_order[k] = k;
	}
	return jh_finish_geometry(g);
}

// Q / ht: tables of the caller instead of the quality-scaled and the default ones (stage level API)
static int encoder_configure(jpeg_hip_encoder *enc, int nframes, int X, int Y, int Nf, int P,
                             const jpeg_hip_params *params, const uint16_t (*Q)[64], const JhHuffSpec (*ht)[4])
{
	if (!enc || !params || nframes <= 0) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(enc->device));
	if (enc->ran) {
		HIPCHK(hipEventSynchronize(enc->done));            // the buffers are about to be reused
	}
	enc->ran = false;
	enc->nframes = 0;                                      // not configured until everything below has succeeded
	if (enc->restart_interval < 0 || enc->restart_interval > 65535) {
		return RET_FAILURE_FILE_UNSUPPORTED;               // DRI holds 16 bits (extension: the reference writes none)
	}
	enc->prm = *params;
	JhGeom &g = enc->g;
	int err = encoder_geometry(X, Y, Nf, P, params, &g);
	if (err) {
		return err;
	}
	memset(enc->Q, 0, sizeof enc->Q);
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			enc->Q[t][i] = 1;
		}
	}
	jh_set_qtable(enc->Q[0], 0, params->q);                // encoder.c:151-152
	jh_set_qtable(enc->Q[1], 1, params->q);
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 4; ++th) {
			jh_default_table(tc, th & 1, &enc->ht[tc][th]);
		}
	}
	if (Q) {
		memcpy(enc->Q, Q, sizeof enc->Q);
	}
	if (ht) {
		memcpy(enc->ht, ht, sizeof enc->ht);
	}

	JhEncodeWork &w = enc->work;
	memset(&w, 0, sizeof w);
	w.nframes = nframes;
	w.nwg = (g.nblk + 255) / 256;
	const size_t raw = (size_t)X * Y * Nf * g.sample_bytes;
	// The unstuffed stream is sized for ordinary images (a little more than the samples themselves); noise at
	// high quality needs more (up to 64 x 27 bits per block): sync() then grows the streams and codes the
	// batch once more (encoder_grow_streams).
	w.us_stride = align_up(raw + (size_t)g.nblk * 4 + 65536, 4096);
	w.out_stride = align_up(4096 + w.us_stride + w.us_stride / 8, 4096);
	w.restart_interval = enc->restart_interval;
	w.nivl = w.restart_interval > 0 ? (g.nmcu + w.restart_interval - 1) / w.restart_interval : 0;

	// Everything below is sized by (frames, geometry, restart interval): a configure() that changes only the
	// quality, the table mode or nothing at all (a service that reconfigures per batch) keeps the buffers.
	const long long sig[8] = {nframes, X, Y, Nf, P, g.H[0], g.V[0], w.restart_interval};
	const bool keep = enc->d_out != nullptr && memcmp(sig, enc->alloc_sig, sizeof sig) == 0;
	if (keep && enc->grown_us) {
		w.us_stride = enc->grown_us;
		w.out_stride = enc->grown_out;
	}
	w.nchunk = (int)(w.us_stride / 4096);
	if (w.us_stride > 0x1ffffff0ull || w.out_stride > 0xfffffff0ull) {
		return RET_FAILURE_OVERFLOW_ERROR;                 // bit positions and byte offsets of a frame are 32-bit
	}
	if (!keep) {
	enc->release();
	enc->grown_us = enc->grown_out = 0;
	memset(enc->alloc_sig, 0, sizeof enc->alloc_sig);      // set again once every buffer is there
	HIPCHK(hipMalloc((void **)&enc->d_quant, nframes * sizeof(JhQuant)));
	HIPCHK(hipMalloc((void **)&enc->d_coefs, (size_t)nframes * g.nblk * 64 * sizeof(int16_t)));
	HIPCHK(hipMalloc((void **)&enc->d_codes, (size_t)nframes * 4 * sizeof(JhHuffEnc)));
	HIPCHK(hipMalloc((void **)&enc->d_runtab, (size_t)nframes * sizeof(JhRunTab)));
	HIPCHK(hipMalloc((void **)&enc->d_hist, (size_t)nframes * 4 * 257 * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_wg_bits, (size_t)nframes * (w.nwg + 1) * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_us, (size_t)nframes * w.us_stride));
	HIPCHK(hipMalloc((void **)&enc->d_chunk_ff, (size_t)nframes * w.nchunk * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_hdr_len, nframes * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_bitlen, (size_t)nframes * g.nblk * sizeof(uint16_t)));
	HIPCHK(hipMalloc((void **)&enc->d_nzmap, (size_t)nframes * g.nblk * sizeof(unsigned long long)));
	HIPCHK(hipMalloc((void **)&enc->d_acdc, (size_t)nframes * g.nblk * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_out, (size_t)nframes * w.out_stride + 16));   // (+16: k_enc_stream_copy reads whole dwords)
	HIPCHK(hipMalloc((void **)&enc->d_state, nframes * sizeof(JhEncState)));
	HIPCHK(hipMalloc((void **)&enc->d_soff, ((size_t)nframes + 1) * sizeof(unsigned long long)));
	HIPCHK(hipMalloc((void **)&enc->d_slen, (size_t)nframes * sizeof(uint32_t)));
	HIPCHK(hipHostMalloc((void **)&enc->h_state, nframes * sizeof(JhEncState), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&enc->h_hdr_len, nframes * sizeof(uint32_t), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&enc->h_slen, nframes * sizeof(uint32_t), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&enc->h_soff, ((size_t)nframes + 1) * sizeof(unsigned long long), hipHostMallocDefault));
	if (w.nivl > 0) {
		HIPCHK(hipMalloc((void **)&enc->d_ivl_raw, (size_t)nframes * w.nivl * sizeof(uint32_t)));
		HIPCHK(hipMalloc((void **)&enc->d_ivl_padsum, (size_t)nframes * (w.nivl + 1) * sizeof(uint32_t)));
		HIPCHK(hipMalloc((void **)&enc->d_ivl_eb, (size_t)nframes * w.nivl * sizeof(uint32_t)));
	}
	HIPCHK(hipMalloc((void **)&enc->d_hdr_parts, 4096));
	HIPCHK(hipMalloc((void **)&enc->d_specs, (size_t)nframes * 4 * sizeof(JhDevSpec)));
	memcpy(enc->alloc_sig, sig, sizeof sig);
	}
	{
		// the header bytes in front of the DHT segments and behind them do not depend on the tables
		std::vector<uint8_t> full;
		jh_produce_headers(g, enc->Q, enc->ht, enc->restart_interval, &full);
		size_t pre = 0, suf = 0;
		if (!split_headers(full, &pre, &suf)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		HIPCHK(hipMemcpy(enc->d_hdr_parts, full.data(), pre, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(enc->d_hdr_parts + pre, full.data() + suf, full.size() - suf, hipMemcpyHostToDevice));
		w.hdr_pre = enc->d_hdr_parts;
		w.hdr_pre_len = (int)pre;
		w.hdr_suf = enc->d_hdr_parts + pre;
		w.hdr_suf_len = (int)(full.size() - suf);
		w.ntab = g.Nf > 1 ? 2 : 1;
		HIPCHK(hipMemset(enc->d_specs, 0, (size_t)nframes * 4 * sizeof(JhDevSpec)));
		w.specs = enc->d_specs;
	}
	w.ivl_raw = enc->d_ivl_raw;
	w.ivl_padsum = enc->d_ivl_padsum;
	w.ivl_eb = enc->d_ivl_eb;
	w.coefs = enc->d_coefs;
	w.codes = enc->d_codes;
	w.hist = enc->d_hist;
	w.bitlen = enc->d_bitlen;
	w.nzmap = enc->d_nzmap;
	w.acdc = enc->d_acdc;
	w.wg_bits = enc->d_wg_bits;
	w.us = enc->d_us;
	w.chunk_ff = enc->d_chunk_ff;
	w.out = enc->d_out;
	w.hdr_len = enc->d_hdr_len;
	w.state = enc->d_state;

	std::vector<JhQuant> quant(nframes);
	for (int f = 0; f < nframes; ++f) {
		memcpy(quant[f].q, enc->Q, sizeof enc->Q);         // raster order: divisors are applied in raster order
	}
	HIPCHK(hipMemcpy(enc->d_quant, quant.data(), nframes * sizeof(JhQuant), hipMemcpyHostToDevice));
	enc->headers.assign(nframes, std::vector<uint8_t>());
	enc->hdr_len.assign(nframes, 0);
	enc->state.assign(nframes, JhEncState());
	// default tables + headers for every frame (the optimised path overwrites them per run)
	err = upload_codes_and_headers(enc, 0, nframes, nullptr);
	if (!err) {
		enc->nframes = nframes;
	}
	return err;
}

extern "C" int jpeg_hip_encoder_configure(jpeg_hip_encoder *enc, int nframes, int X, int Y, int Nf, int P,
                                          const jpeg_hip_params *params)
{
	return encoder_configure(enc, nframes, X, Y, Nf, P, params, nullptr, nullptr);
}

extern "C" int jpeg_hip_encoder_info(const jpeg_hip_encoder *enc, jpeg_hip_frame_info *info)
{
	if (!enc || enc->nframes == 0) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	fill_info(enc->g, info);
	return RET_SUCCESS;
}

// What sync() reads comes to page-locked memory behind the run, on its stream, and an event marks the end: sync()
// waits for that event only (not for what the caller queued behind the run) and touches no stream.
// stream_dst != NULL: the codestreams leave packed (k_enc_stream_*) in front of that.
static int encoder_queue_results(jpeg_hip_encoder *enc, hipStream_t s, const JhTimerHook *th)
{
	const int n = enc->nframes;
	if (enc->stream_dst) {
		if (jh_launch_enc_stream(enc->work, enc->stream_dst, enc->stream_cap, enc->d_soff, enc->d_slen, s, th)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		HIPCHK(hipMemcpyAsync(enc->h_soff, enc->d_soff, ((size_t)n + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
		HIPCHK(hipMemcpyAsync(enc->h_slen, enc->d_slen, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	}
	HIPCHK(hipMemcpyAsync(enc->h_state, enc->d_state, n * sizeof(JhEncState), hipMemcpyDeviceToHost, s));
	if (enc->prm.optimize) {
		HIPCHK(hipMemcpyAsync(enc->h_hdr_len, enc->d_hdr_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	}
	HIPCHK(hipEventRecord(enc->done, s));
	return RET_SUCCESS;
}

static int encoder_run(jpeg_hip_encoder *enc, const void *d_samples, size_t frame_stride, void *stream, void *dst, size_t cap)
{
	if (!enc || enc->nframes == 0 || !d_samples) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	const JhGeom &g = enc->g;
	if (frame_stride < (size_t)g.X * g.Y * g.Nf * g.sample_bytes) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(enc->device));
	hipStream_t s = (hipStream_t)stream;
	if (enc->ran) {
		// the page-locked copies of the last run: read by sync() -- a caller that skipped it must not find them half overwritten
		HIPCHK(hipEventSynchronize(enc->done));
	}
	enc->last_stream = s;
	enc->stream_dst = dst;
	enc->stream_cap = cap;
	enc->timer.reset();
	const JhEncodeWork &w = enc->work;
	HIPCHK(hipMemsetAsync(enc->d_state, 0, enc->nframes * sizeof(JhEncState), s));
	const JhTimerHook th = enc->timer.hook();
	enc->timer.begin("k_encode_pixels", s);
	// default tables: the codes are known, and the pixel kernel counts every block's AC bits while it holds the block
	const bool acbits = !enc->prm.optimize;
	if (jh_launch_encode_pixels(g, enc->nframes, d_samples, frame_stride, enc->d_quant, enc->d_coefs, enc->d_nzmap, enc->d_acdc,
	                            acbits ? enc->d_codes : nullptr, enc->d_runtab, enc->d_state, s, enc->float_tap)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	enc->timer.end(s);
	if (enc->prm.optimize) {
		// write_ecs_dry() (encoder.c:525-558): symbol statistics, then adapt_huffman_table (huffman.c:508-537)
		// and the headers that carry the new tables -- all on the device, nothing waits for the host
		if (jh_launch_enc_histogram(w, g, s, &th) || jh_launch_enc_adapt(w, g, s, &th)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	}
	if (jh_launch_enc_entropy(w, g, s, &th, acbits)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	enc->acbits_ready = acbits;
	int err = encoder_queue_results(enc, s, &th);
	if (err) {
		return err;
	}
	enc->ran = true;
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_encoder_run(jpeg_hip_encoder *enc, const void *d_samples, size_t frame_stride, void *stream)
{
	if (enc) {
		enc->host_dst = nullptr;
	}
	return encoder_run(enc, d_samples, frame_stride, stream, nullptr, 0);
}

// Motion JPEG stream out (the encode direction): see include/jpeg_amd.h
extern "C" int jpeg_hip_encoder_run_stream(jpeg_hip_encoder *enc, const void *d_samples, size_t frame_stride,
                                           void *out, size_t cap, void *stream)
{
	if (!enc || !out) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(enc->device));
	// the kernels store through `out`: it must be memory this device can write -- device memory, or page-locked host
	// memory (a pointer into pageable memory would end in a memory access fault, not in an error code)
	hipPointerAttribute_t at;
	memset(&at, 0, sizeof at);
	if (hipPointerGetAttributes(&at, out) != hipSuccess ||
	    (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged)) {
		(void)hipGetLastError();
		fprintf(stderr, "jpeg_amd: jpeg_hip_encoder_run_stream: `out` is neither device memory nor page-locked host memory "
		                "(jpeg_hip_host_alloc)\n");
		return RET_FAILURE_LOGIC_ERROR;
	}
	// Device memory: the kernels pack straight into it.  Host memory: they pack into a staging buffer in HBM, and
	// sync_stream() -- which knows the stream's length by then -- sends exactly that many bytes over with ONE DMA.
	// (Kernels that store across the link directly reach its rate too, 53 GB/s, but while they run the device runs
	// nothing else: posted writes fill the fabric's queues and every other kernel's stores wait behind them --
	// measured, three encoders on three queues took 3 x (2.9 + 2.3) ms per round.  The DMA engines do not.)
	void *dev_view = out;
	enc->host_dst = nullptr;
	if (at.type == hipMemoryTypeHost) {
		if (enc->ran) {
			HIPCHK(hipEventSynchronize(enc->done));            // the staging buffer may be replaced
		}
		if (enc->d_pack_cap < cap) {
			hipFree(enc->d_pack);
			enc->d_pack = nullptr;
			enc->d_pack_cap = 0;
			HIPCHK(hipMalloc((void **)&enc->d_pack, cap ? cap : 1));
			enc->d_pack_cap = cap;
		}
		if (!enc->copy_stream) {
			HIPCHK(hipStreamCreateWithFlags(&enc->copy_stream, hipStreamNonBlocking));
			HIPCHK(hipEventCreateWithFlags(&enc->copied, hipEventDisableTiming));
		}
		dev_view = enc->d_pack;
	}
	int err = encoder_run(enc, d_samples, frame_stride, stream, dev_view, cap);
	enc->host_dst = !err && at.type == hipMemoryTypeHost ? out : nullptr;
	return err;
}

// A frame's entropy-coded data did not fit the unstuffed stream or the output slot (noise at high quality: the
// streams are sized for ordinary images).  The reference writes to a FILE* without limit, so: grow both to
// what the device counted (the bit total is exact even when it does not fit), worst case for the stuffing,
// and code the batch once more from the coefficients that are still in HBM.
static int encoder_grow_streams(jpeg_hip_encoder *enc)
{
	JhEncodeWork &w = enc->work;
	const int n = enc->nframes;
	uint32_t max_bits = 0;
	for (int f = 0; f < n; ++f) {
		max_bits = enc->state[f].total_bits > max_bits ? enc->state[f].total_bits : max_bits;
	}
	size_t us = align_up((size_t)max_bits / 8 + (size_t)w.nivl + 65536, 4096);
	us = us > w.us_stride ? us : w.us_stride;
	const size_t out = align_up(4096 + 2 * us + 2 * (size_t)w.nivl + 16, 4096);
	if (us > 0x1ffffff0ull || out > 0xfffffff0ull || (us == w.us_stride && out <= w.out_stride)) {
		return RET_FAILURE_OVERFLOW_ERROR;
	}
	hipFree(enc->d_us);
	hipFree(enc->d_out);
	hipFree(enc->d_chunk_ff);
	enc->d_us = nullptr;
	enc->d_chunk_ff = nullptr;
	enc->d_out = nullptr;
	memset(enc->alloc_sig, 0, sizeof enc->alloc_sig);      // a failure below leaves an encoder that configure() rebuilds
	w.us_stride = us;
	w.out_stride = out;
	w.nchunk = (int)(us / 4096);
	HIPCHK(hipMalloc((void **)&enc->d_us, (size_t)n * w.us_stride));
	HIPCHK(hipMalloc((void **)&enc->d_chunk_ff, (size_t)n * w.nchunk * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&enc->d_out, (size_t)n * w.out_stride + 16));
	w.us = enc->d_us;
	w.chunk_ff = enc->d_chunk_ff;
	w.out = enc->d_out;
	enc->grown_us = us;
	enc->grown_out = out;
	const JhGeom &g = enc->g;
	const long long sig[8] = {n, g.X, g.Y, g.Nf, g.P, g.H[0], g.V[0], w.restart_interval};
	memcpy(enc->alloc_sig, sig, sizeof sig);
	// the state records keep what the pixel and statistics stages found; the stream totals start again
	for (int f = 0; f < n; ++f) {
		enc->state[f].status &= ~(uint32_t)JH_EST_OVERFLOW;
		enc->state[f].total_bits = enc->state[f].us_bytes = enc->state[f].ecs_bytes = 0;
	}
	hipStream_t s = enc->last_stream;
	HIPCHK(hipMemcpy(enc->d_state, enc->state.data(), n * sizeof(JhEncState), hipMemcpyHostToDevice));
	if (enc->prm.optimize) {
		if (jh_launch_enc_adapt(w, g, s, nullptr)) {       // the statistics are still there: tables and headers again
			return RET_FAILURE_LOGIC_ERROR;
		}
	} else {
		int err = upload_codes_and_headers(enc, 0, n, s);
		if (err) {
			return err;
		}
	}
	if (jh_launch_enc_entropy(w, g, s, nullptr, enc->acbits_ready)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	return encoder_queue_results(enc, s, nullptr);
}

static int encoder_sync(jpeg_hip_encoder *enc, int *frame_status, size_t *jpg_len, size_t *frame_off);

extern "C" int jpeg_hip_encoder_sync(jpeg_hip_encoder *enc, int *frame_status, size_t *jpg_len)
{
	return encoder_sync(enc, frame_status, jpg_len, nullptr);
}

extern "C" int jpeg_hip_encoder_sync_stream(jpeg_hip_encoder *enc, int *frame_status, size_t *frame_off)
{
	if (!enc || !enc->ran || !enc->stream_dst || !frame_off) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	return encoder_sync(enc, frame_status, nullptr, frame_off);
}

// frame_off != NULL: the run was a run_stream(); frame_off[nframes + 1] receives where each frame's codestream starts in
// the packed stream (a failed frame takes no room: frame_off[f + 1] == frame_off[f])
static int encoder_sync(jpeg_hip_encoder *enc, int *frame_status, size_t *jpg_len, size_t *frame_off)
{
	if (!enc || !enc->ran) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	for (int f = 0; f < enc->nframes; ++f) {               // what every early return below leaves behind
		if (frame_status) {
			frame_status[f] = RET_FAILURE_LOGIC_ERROR;
		}
		if (jpg_len) {
			jpg_len[f] = 0;
		}
	}
	HIPCHK(hipSetDevice(enc->device));
	for (int attempt = 0;; ++attempt) {
		HIPCHK(hipEventSynchronize(enc->done));            // this run, not what the caller queued behind it
		memcpy(enc->state.data(), enc->h_state, enc->nframes * sizeof(JhEncState));   // run() brought them over
		bool over = false;
		for (int f = 0; f < enc->nframes; ++f) {
			const JhEncState &st = enc->state[f];
			over = over || (st.first_err == 0 && !(st.status & (JH_EST_DC_RANGE | JH_EST_NO_CODE)) && (st.status & JH_EST_OVERFLOW));
		}
		if (!over || attempt > 0) {
			break;
		}
		const int err = encoder_grow_streams(enc);
		if (err == RET_FAILURE_OVERFLOW_ERROR) {
			break;                                         // beyond what a frame may hold: reported per frame below
		}
		if (err) {
			return err;
		}
	}
	if (enc->prm.optimize) {
		memcpy(enc->hdr_len.data(), enc->h_hdr_len, enc->nframes * sizeof(uint32_t));
	}
	int worst = RET_SUCCESS;
	for (int f = 0; f < enc->nframes; ++f) {
		const JhEncState &st = enc->state[f];
		int r = RET_SUCCESS;
		if (st.first_err != 0) {
			// the error of the first failing block in scan order, as the serial encoder meets it
			r = ((~st.first_err) & 1u) ? JPEG_HIP_ERR_NO_CODE : JPEG_HIP_ERR_DC_RANGE;
		} else if (st.status & JH_EST_DC_RANGE) {
			r = JPEG_HIP_ERR_DC_RANGE;
		} else if (st.status & JH_EST_NO_CODE) {
			r = JPEG_HIP_ERR_NO_CODE;
		} else if (st.status & JH_EST_OVERFLOW) {
			r = RET_FAILURE_OVERFLOW_ERROR;
		} else if (frame_off && enc->h_slen[f] == JH_STREAM_NOFIT) {
			r = RET_FAILURE_OVERFLOW_ERROR;                    // a complete codestream, and no room left for it in `out`
		}
		if (frame_status) {
			frame_status[f] = r;
		}
		if (jpg_len) {
			jpg_len[f] = r ? 0 : (size_t)enc->hdr_len[f] + st.ecs_bytes + 2;
		}
		if (frame_off) {
			frame_off[f] = (size_t)enc->h_soff[f];
			if (f == enc->nframes - 1) {
				frame_off[f + 1] = (size_t)enc->h_soff[f + 1];
			}
		}
		if (r && !worst) {
			worst = r;
		}
	}
	if (frame_off && enc->host_dst && enc->h_soff[enc->nframes] > 0) {
		// the packed stream, exactly as long as it is, from the staging buffer to the caller's page-locked memory
		HIPCHK(hipMemcpyAsync(enc->host_dst, enc->d_pack, (size_t)enc->h_soff[enc->nframes], hipMemcpyDeviceToHost, enc->copy_stream));
		HIPCHK(hipEventRecord(enc->copied, enc->copy_stream));
		HIPCHK(hipEventSynchronize(enc->copied));
	}
	return worst;
}

extern "C" int jpeg_hip_encoder_get_jpeg(jpeg_hip_encoder *enc, int frame, uint8_t *out, size_t cap, size_t *len)
{
	if (!enc || !enc->ran || frame < 0 || frame >= enc->nframes) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	const size_t n = (size_t)enc->hdr_len[frame] + enc->state[frame].ecs_bytes + 2;
	*len = n;
	if (cap < n) {
		return RET_FAILURE_OVERFLOW_ERROR;
	}
	HIPCHK(hipSetDevice(enc->device));
	HIPCHK(hipMemcpy(out, enc->d_out + (size_t)frame * enc->work.out_stride, n, hipMemcpyDeviceToHost));
	return RET_SUCCESS;
}

extern "C" const void *jpeg_hip_encoder_device_jpeg(jpeg_hip_encoder *enc, int frame, size_t *len)
{
	if (!enc || !enc->ran || frame < 0 || frame >= enc->nframes) {
		return nullptr;
	}
	*len = (size_t)enc->hdr_len[frame] + enc->state[frame].ecs_bytes + 2;
	return enc->d_out + (size_t)frame * enc->work.out_stride;
}

extern "C" int jpeg_hip_encoder_get_coefficients(jpeg_hip_encoder *enc, int frame, int k, int32_t *out)
{
	if (!enc || !enc->ran || frame < 0 || frame >= enc->nframes || k < 0 || k >= enc->g.ncomp) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipSetDevice(enc->device));
	const size_t n = (size_t)enc->g.nblk * 64;
	std::vector<int16_t> fc(n);
	HIPCHK(hipMemcpy(fc.data(), enc->d_coefs + (size_t)frame * n, n * sizeof(int16_t), hipMemcpyDeviceToHost));
	coefs_to_reference_layout(enc->g, fc.data(), k, out);
	return RET_SUCCESS;
}

extern "C" void jpeg_hip_encoder_set_float_tap(jpeg_hip_encoder *enc, void *d_floats)
{
	if (enc) {
		enc->float_tap = (float *)d_floats;
	}
}

extern "C" void jpeg_hip_encoder_set_timing(jpeg_hip_encoder *enc, int on) { enc->timer.on = on != 0; }

extern "C" int jpeg_hip_encoder_get_timing(jpeg_hip_encoder *enc, const char **names, float *ms, int cap)
{
	hipSetDevice(enc->device);
	return enc->timer.collect(names, ms, cap);
}

// ======================================================================================= the stages one by one
//
// dequantize / quantize / inverse_dct / forward_dct (imgproc.h:8-16) and frame_to_rgb / frame_to_ycc (frame.h:21,33) on the
// reference's own buffers, one kernel each: host pointers in and out, synchronous, like the other stage-level calls.
// integration/ref_glue.c (-DGLUE_STAGES) puts them behind the reference's function names.

struct StageBuf {                      // device copies of a stage call's arrays, freed when the call returns
	void *p[3] = {nullptr, nullptr, nullptr};
	~StageBuf()
	{
		for (void *q : p) {
			hipFree(q);
		}
	}
};

static int stage_blocks(int device, int what, size_t nblocks, const void *in, size_t in_elem, const uint16_t *Q,
                        void *out, size_t out_elem, int P)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if ((!in || !out) && nblocks) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (nblocks == 0) {
		return RET_SUCCESS;
	}
	if (P < 1 || P > 16) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	const size_t n = nblocks * 64;
	StageBuf b;
	const bool inplace = what == JH_STAGE_IDCT || what == JH_STAGE_FDCT;
	HIPCHK(hipMalloc(&b.p[0], n * in_elem));
	HIPCHK(hipMemcpy(b.p[0], in, n * in_elem, hipMemcpyHostToDevice));
	if (!inplace) {
		if (!Q) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		HIPCHK(hipMalloc(&b.p[1], 64 * sizeof(uint16_t)));
		HIPCHK(hipMemcpy(b.p[1], Q, 64 * sizeof(uint16_t), hipMemcpyHostToDevice));
		HIPCHK(hipMalloc(&b.p[2], n * out_elem));
	}
	if (jh_launch_stage(what, b.p[0], b.p[1], b.p[2], inplace ? nblocks : n, P, 0, 0, 0, nullptr)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipMemcpy(out, inplace ? b.p[0] : b.p[2], n * out_elem, hipMemcpyDeviceToHost));
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_dequantize_blocks(int device, size_t nblocks, const int32_t *coef, const uint16_t Q[64], float *out)
{
	return stage_blocks(device, JH_STAGE_DEQUANTIZE, nblocks, coef, sizeof(int32_t), Q, out, sizeof(float), 8);
}

extern "C" int jpeg_hip_quantize_blocks(int device, size_t nblocks, const float *in, const uint16_t Q[64], int32_t *out)
{
	if (Q) {
		for (int i = 0; i < 64; ++i) {
			if (Q[i] == 0) {
				return RET_FAILURE_FILE_UNSUPPORTED;           // quantize_block() would divide by zero
			}
		}
	}
	return stage_blocks(device, JH_STAGE_QUANTIZE, nblocks, in, sizeof(float), Q, out, sizeof(int32_t), 8);
}

extern "C" int jpeg_hip_inverse_dct_blocks(int device, size_t nblocks, float *blocks, int P)
{
	return stage_blocks(device, JH_STAGE_IDCT, nblocks, blocks, sizeof(float), nullptr, blocks, sizeof(float), P);
}

extern "C" int jpeg_hip_forward_dct_blocks(int device, size_t nblocks, float *blocks, int P)
{
	return stage_blocks(device, JH_STAGE_FDCT, nblocks, blocks, sizeof(float), nullptr, blocks, sizeof(float), P);
}

static int stage_colour(int device, int what, float *data, int components, int X, int Y, size_t size_x, size_t size_y, int P)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (components == 1) {
		return RET_SUCCESS;                                    // nothing to do (frame.c:176-178, 236-238)
	}
	const bool ok = what == JH_STAGE_TO_RGB ? (components == 3 || components == 4) : components == 3;
	if (!data || !ok || X <= 0 || Y <= 0 || (size_t)X > size_x || (size_t)Y > size_y || P < 1 || P > 16) {
		return RET_FAILURE_LOGIC_ERROR;                        // (the reference abort()s on other component counts)
	}
	const size_t bytes = size_x * size_y * (size_t)components * sizeof(float);
	StageBuf b;
	HIPCHK(hipMalloc(&b.p[0], bytes));
	HIPCHK(hipMemcpy(b.p[0], data, bytes, hipMemcpyHostToDevice));
	if (jh_launch_stage(what, b.p[0], (const void *)(size_t)P, nullptr, (size_t)X * Y, components, X, Y, size_x, nullptr)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	HIPCHK(hipMemcpy(data, b.p[0], bytes, hipMemcpyDeviceToHost));
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_frame_to_rgb(int device, float *data, int components, int X, int Y, size_t size_x, size_t size_y, int P)
{
	return stage_colour(device, JH_STAGE_TO_RGB, data, components, X, Y, size_x, size_y, P);
}

extern "C" int jpeg_hip_frame_to_ycc(int device, float *data, int components, int X, int Y, size_t size_x, size_t size_y, int P)
{
	return stage_colour(device, JH_STAGE_TO_YCC, data, components, X, Y, size_x, size_y, P);
}

// ======================================================================================= self-test

extern "C" int jpeg_hip_selftest_divide(int device, int ties_per_divisor, unsigned seed, unsigned long long out[3])
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!out || ties_per_divisor < 1) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	unsigned long long *d = nullptr;
	HIPCHK(hipMalloc((void **)&d, 3 * sizeof *d));
	hipError_t e = hipMemset(d, 0, 3 * sizeof *d);
	if (e == hipSuccess && jh_launch_selftest_divide(d, ties_per_divisor, seed, nullptr)) {
		e = hipErrorUnknown;
	}
	e = e == hipSuccess ? hipMemcpy(out, d, 3 * sizeof *d, hipMemcpyDeviceToHost) : e;
	hipFree(d);
	return e == hipSuccess ? RET_SUCCESS : RET_FAILURE_LOGIC_ERROR;
}

extern "C" int jpeg_hip_selftest_ycc8(int device, unsigned long long out[3])
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!out) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	unsigned long long *d = nullptr;
	HIPCHK(hipMalloc((void **)&d, 3 * sizeof *d));
	hipError_t e = hipMemset(d, 0, 3 * sizeof *d);
	if (e == hipSuccess && jh_launch_selftest_ycc8(d, nullptr)) {
		e = hipErrorUnknown;
	}
	e = e == hipSuccess ? hipMemcpy(out, d, 3 * sizeof *d, hipMemcpyDeviceToHost) : e;
	hipFree(d);
	return e == hipSuccess ? RET_SUCCESS : RET_FAILURE_LOGIC_ERROR;
}

// ======================================================================================= no-allocation entry point

// jpeg_hip_encode_noalloc(): jpeg_hip_encoder_run() for ONE frame without any device allocation.  No
// counterpart in the reference; for callers that cannot call hipMalloc (HIP graph capture, tight serving
// loops, harnesses that forbid it).  Everything the encoder kernels need beyond the samples and the
// codestream -- coefficients, bit lengths, prefix sums, the unstuffed stream, histograms, tables, the state
// record -- lives in ONE file-scope device array of a fixed, modest size.  Every GPU the library is used on
// holds its own copy of the array (at its own address) until the process ends; every call on a GPU shares
// that GPU's array, so calls that overlap on one GPU race.
#define JH_NOALLOC_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scratch[JH_NOALLOC_BYTES];

// `scratch` above is, in host code, only a handle for hipGetSymbolAddress(): the device address is looked up
// once per device (with that device current) and kept here; host code uses these pointers and nothing else.
#define JH_NOALLOC_MAX_DEVICES 64
static unsigned char *g_scratch_ptr[JH_NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_of_device(int device, unsigned char **ptr)
{
	if (device < 0 || device >= JH_NOALLOC_MAX_DEVICES) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	std::lock_guard<std::mutex> lock(g_scratch_mu);
	if (!g_scratch_ptr[device]) {
		void *p = nullptr;
		HIPCHK(hipSetDevice(device));
		HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
		if (!p) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		g_scratch_ptr[device] = (unsigned char *)p;
	}
	*ptr = g_scratch_ptr[device];
	return RET_SUCCESS;
}

// what the set-up kernel receives BY VALUE (kernel arguments: no host-to-device copy, no staging buffer)
struct NoAllocSetup {
	JhQuant quant;                    // raster order, as jh_launch_encode_pixels() wants it
	uint8_t hdr[512];                 // [0, pre_len): SOI DQT SOF0; [pre_len, pre_len + suf_len): SOS
	int pre_len, suf_len;
	uint8_t L[4][16];                 // the default tables [Tc * 2 + Th] (mjpeg.h), as a DHT segment carries them
	uint8_t V[4][256];
	// where things live inside the scratch array
	uint32_t off_quant, off_state, off_hdr_len, off_hdr, off_specs, off_codes;
};

__global__ __launch_bounds__(256) void k_noalloc_setup(const NoAllocSetup a, unsigned char *base)
{
	const int tid = threadIdx.x;
	reinterpret_cast<uint16_t *>(base + a.off_quant)[tid] = (&a.quant.q[0][0])[tid];   // 4 x 64 entries
	for (int i = tid; i < a.pre_len + a.suf_len; i += 256) {
		(base + a.off_hdr)[i] = a.hdr[i];
	}
	if (tid < (int)(sizeof(JhEncState) / 4)) {
		reinterpret_cast<uint32_t *>(base + a.off_state)[tid] = 0;
	}
	JhDevSpec *specs = reinterpret_cast<JhDevSpec *>(base + a.off_specs);
	JhHuffEnc *codes = reinterpret_cast<JhHuffEnc *>(base + a.off_codes);
	for (int t = 0; t < 4; ++t) {
		specs[t].V[tid] = a.V[t][tid];
		codes[t].code[tid] = 0;
		if (tid < 16) {
			specs[t].L[tid] = a.L[t][tid];
		}
	}
	__syncthreads();
	if (tid < 4) {
		// Annex C (huffman.c:45-148): canonical codes in table order; a symbol listed twice keeps its first
		// code (value_to_vlc, huffman.c:238-245)
		uint32_t code = 0;
		int p = 0;
		for (int len = 1; len <= 16; ++len) {
			for (int i = 0; i < a.L[tid][len - 1] && p < 256; ++i, ++p) {
				const int sym = a.V[tid][p];
				if (codes[tid].code[sym] == 0) {
					codes[tid].code[sym] = (code << 8) | (uint32_t)len;
				}
				code++;
			}
			code <<= 1;
		}
		specs[tid].mt = (uint32_t)p;
	}
}

// {RET code, codestream length} for the caller, on the device
__global__ void k_noalloc_result(const JhEncState *st, const uint32_t *hdr_len, uint32_t *result)
{
	uint32_t r = RET_SUCCESS;
	if (st->first_err != 0) {
		r = ((~st->first_err) & 1u) ? (uint32_t)JPEG_HIP_ERR_NO_CODE : (uint32_t)JPEG_HIP_ERR_DC_RANGE;
	} else if (st->status & JH_EST_DC_RANGE) {
		r = (uint32_t)JPEG_HIP_ERR_DC_RANGE;
	} else if (st->status & JH_EST_NO_CODE) {
		r = (uint32_t)JPEG_HIP_ERR_NO_CODE;
	} else if (st->status & JH_EST_OVERFLOW) {
		r = RET_FAILURE_OVERFLOW_ERROR;
	}
	result[0] = r;
	result[1] = r ? 0u : hdr_len[0] + st->ecs_bytes + 2u;
}

extern "C" size_t jpeg_hip_encode_noalloc_limit(void) { return JH_NOALLOC_BYTES; }

// Set-up, once per device (optional: the entry point does it on its first call on a device otherwise).  The
// HIP runtime loads this library's code object on the device at this point, and the fixed array comes into
// being with it: a caller that must not see device memory change inside jpeg_hip_encode_noalloc() calls this
// first, outside its no-allocation region.
extern "C" int jpeg_hip_encode_noalloc_init(int device)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	unsigned char *base = nullptr;
	return scratch_of_device(device, &base);
}

extern "C" int jpeg_hip_encode_noalloc(int device, const jpeg_hip_params *params, int X, int Y, int Nf, int P,
                                       const void *d_samples, void *d_jpeg, size_t cap, uint32_t *d_result, void *stream)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!params || !d_samples || !d_jpeg || !d_result) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	JhGeom g;
	err = encoder_geometry(X, Y, Nf, P, params, &g);
	if (err) {
		return err;
	}
	unsigned char *base = nullptr;
	err = scratch_of_device(device, &base);
	if (err) {
		return err;
	}
	if (cap < 8192 || cap > 0xffffffffu) {
		fprintf(stderr, "jpeg_amd: jpeg_hip_encode_noalloc: the codestream buffer must hold 8192 .. 2^32 - 1 bytes (got %zu)\n", cap);
		return RET_FAILURE_OVERFLOW_ERROR;
	}

	// ---- carve the array: fixed-size pieces first, the unstuffed stream takes what is left
	JhEncodeWork w;
	memset(&w, 0, sizeof w);
	w.nframes = 1;
	w.nwg = (g.nblk + 255) / 256;
	size_t off = 0;
	auto take = [&](size_t bytes) {
		const size_t at = off;
		off = align_up(off + bytes, 256);
		return at;
	};
	NoAllocSetup su;
	memset(&su, 0, sizeof su);
	su.off_quant = (uint32_t)take(sizeof(JhQuant));
	su.off_state = (uint32_t)take(sizeof(JhEncState));
	su.off_hdr_len = (uint32_t)take(sizeof(uint32_t));
	su.off_hdr = (uint32_t)take(sizeof su.hdr);
	su.off_specs = (uint32_t)take(4 * sizeof(JhDevSpec));
	su.off_codes = (uint32_t)take(4 * sizeof(JhHuffEnc));
	const size_t off_runtab = take(sizeof(JhRunTab));
	const size_t off_hist = take(4 * 257 * sizeof(uint32_t));
	const size_t off_chunk_ff = take((JH_NOALLOC_BYTES / 4096) * sizeof(uint32_t));
	const size_t off_wg_bits = take((size_t)(w.nwg + 1) * sizeof(uint32_t));
	const size_t off_bitlen = take((size_t)g.nblk * sizeof(uint16_t));
	const size_t off_nzmap = take((size_t)g.nblk * sizeof(unsigned long long));
	const size_t off_acdc = take((size_t)g.nblk * sizeof(uint32_t));
	const size_t off_coefs = take((size_t)g.nblk * 64 * sizeof(int16_t));
	const size_t raw = (size_t)X * Y * Nf * g.sample_bytes;
	const size_t us_need = align_up(raw + (size_t)g.nblk * 4 + 65536, 4096);    // what jpeg_hip_encoder_configure() reserves
	if (off + us_need > JH_NOALLOC_BYTES) {
		fprintf(stderr, "jpeg_amd: jpeg_hip_encode_noalloc: a %dx%d frame of %d component(s) needs %zu bytes of workspace, the "
		                "fixed array holds %u (use the encoder object)\n", X, Y, Nf, off + us_need, JH_NOALLOC_BYTES);
		return RET_FAILURE_OVERFLOW_ERROR;
	}
	const size_t off_us = off;
	w.us_stride = (JH_NOALLOC_BYTES - off) / 4096 * 4096;
	w.nchunk = (int)(w.us_stride / 4096);
	w.out_stride = cap;
	w.coefs = reinterpret_cast<int16_t *>(base + off_coefs);
	w.codes = reinterpret_cast<JhHuffEnc *>(base + su.off_codes);
	w.hist = reinterpret_cast<uint32_t *>(base + off_hist);
	w.bitlen = reinterpret_cast<uint16_t *>(base + off_bitlen);
	w.nzmap = reinterpret_cast<unsigned long long *>(base + off_nzmap);
	w.acdc = reinterpret_cast<uint32_t *>(base + off_acdc);
	w.wg_bits = reinterpret_cast<uint32_t *>(base + off_wg_bits);
	w.us = reinterpret_cast<uint32_t *>(base + off_us);
	w.chunk_ff = reinterpret_cast<uint32_t *>(base + off_chunk_ff);
	w.out = (uint8_t *)d_jpeg;
	w.hdr_len = reinterpret_cast<uint32_t *>(base + su.off_hdr_len);
	w.specs = reinterpret_cast<JhDevSpec *>(base + su.off_specs);
	w.state = reinterpret_cast<JhEncState *>(base + su.off_state);
	w.ntab = Nf > 1 ? 2 : 1;

	// ---- tables and header bytes, handed to the set-up kernel by value
	uint16_t Q[4][64];
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			Q[t][i] = 1;
		}
	}
	jh_set_qtable(Q[0], 0, params->q);                     // encoder.c:151-152
	jh_set_qtable(Q[1], 1, params->q);
	memcpy(su.quant.q, Q, sizeof Q);
	JhHuffSpec ht[2][4];
	for (int tc = 0; tc < 2; ++tc) {
		for (int th = 0; th < 4; ++th) {
			jh_default_table(tc, th & 1, &ht[tc][th]);
		}
		for (int th = 0; th < 2; ++th) {
			memcpy(su.L[tc * 2 + th], ht[tc][th].L, 16);
			memcpy(su.V[tc * 2 + th], ht[tc][th].V, 256);
		}
	}
	std::vector<uint8_t> full;
	jh_produce_headers(g, Q, ht, 0, &full);
	size_t pre = 0, suf = 0;
	if (!split_headers(full, &pre, &suf) || pre + (full.size() - suf) > sizeof su.hdr) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	memcpy(su.hdr, full.data(), pre);
	memcpy(su.hdr + pre, full.data() + suf, full.size() - suf);
	su.pre_len = (int)pre;
	su.suf_len = (int)(full.size() - suf);
	w.hdr_pre = base + su.off_hdr;
	w.hdr_pre_len = su.pre_len;
	w.hdr_suf = base + su.off_hdr + pre;
	w.hdr_suf_len = su.suf_len;

	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(k_noalloc_setup, dim3(1), dim3(256), 0, s, su, base);
	if (hipGetLastError() != hipSuccess) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (jh_launch_encode_pixels(g, 1, d_samples, raw, reinterpret_cast<const JhQuant *>(base + su.off_quant),
	                            reinterpret_cast<int16_t *>(base + off_coefs), w.nzmap, w.acdc, params->optimize ? nullptr : w.codes,
	                            reinterpret_cast<JhRunTab *>(base + off_runtab), w.state, s)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (params->optimize) {
		if (jh_launch_enc_histogram(w, g, s, nullptr) || jh_launch_enc_adapt(w, g, s, nullptr)) {
			return RET_FAILURE_LOGIC_ERROR;
		}
	} else if (jh_launch_enc_headers(w, g, s)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (jh_launch_enc_entropy(w, g, s, nullptr, !params->optimize)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	hipLaunchKernelGGL(k_noalloc_result, dim3(1), dim3(1), 0, s, w.state, w.hdr_len, d_result);
	return hipGetLastError() == hipSuccess ? RET_SUCCESS : RET_FAILURE_LOGIC_ERROR;
}

// ======================================================================================= single-image drop-ins

// ---- stage level: the four hot loops on flattened `struct context` members (jpeg_amd.h) --------------

// the reference's int_buffer layout of component k -> scan-order / zig-zag int16 coefficients of one frame
static bool reference_layout_to_coefs(const JhGeom &g, int k, const int32_t *in, int16_t *fc)
{
	bool fits = true;
	for (int by = 0; by < g.by[k]; ++by) {
		for (int bx = 0; bx < g.bx[k]; ++bx) {
			const int mx = bx / g.H[k], h = bx % g.H[k], my = by / g.V[k], v = by % g.V[k];
			const size_t blk = (size_t)(my * g.m_x + mx) * g.B + g.comp_first[k] + v * g.H[k] + h;
			const int32_t *src = in + ((size_t)by * g.bx[k] + bx) * 64;
			for (int z = 0; z < 64; ++z) {
				const int32_t c = src[jh_zigzag[z]];
				fits = fits && c >= -32768 && c <= 32767;
				fc[blk * 64 + z] = (int16_t)c;
			}
		}
	}
	return fits;
}

static int geom_from_desc(const jpeg_hip_frame_desc *fd, JhGeom *g)
{
	memset(g, 0, sizeof *g);
	if (!fd || fd->Nf < 1 || fd->Nf > JH_MAX_COMP || fd->Nf == 2 || fd->X <= 0 || fd->Y <= 0 || fd->X > 65535 || fd->Y > 65535) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	g->P = fd->P; g->X = fd->X; g->Y = fd->Y; g->Nf = fd->Nf; g->ncomp = fd->Nf;
	for (int k = 0; k < fd->Nf; ++k) {
		if (fd->Tq[k] < 0 || fd->Tq[k] > 3 || fd->H[k] < 1 || fd->H[k] > 4 || fd->V[k] < 1 || fd->V[k] > 4) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		g->id[k] = k + 1;
		g->H[k] = fd->H[k];
		g->V[k] = fd->V[k];
		g->Tq[k] = fd->Tq[k];
		g->scan_order[k] = k;
	}
	return RET_SUCCESS;
}

static void spec_from_htable(const jpeg_hip_htable *t, JhHuffSpec *s)
{
	memcpy(s->L, t->L, 16);
	memcpy(s->V, t->V, 256);
}

// The stage-level calls below work on host memory, one image per call.  Their device objects and buffers are kept
// per device between calls and only grow (hipMalloc / hipFree wait for every stream of the device); a call holds
// the device's set for its duration, so calls on one device take turns.  Held until the process ends.
struct StageObjects {
	std::mutex m;
	jpeg_hip_decoder *dec = nullptr;
	jpeg_hip_encoder *enc = nullptr;
	void *buf[3] = {nullptr, nullptr, nullptr};
	size_t cap[3] = {0, 0, 0};
	void *buffer(int slot, size_t bytes)
	{
		if (cap[slot] < bytes || !buf[slot]) {
			if (buf[slot]) {
				hipFree(buf[slot]);
				buf[slot] = nullptr;
				cap[slot] = 0;
			}
			const size_t want = bytes + bytes / 4 + 256;
			if (hipMalloc(&buf[slot], want) != hipSuccess) {
				buf[slot] = nullptr;
				return nullptr;
			}
			cap[slot] = want;
		}
		return buf[slot];
	}
	int decoder(int device, jpeg_hip_decoder **out)
	{
		const int err = dec ? RET_SUCCESS : jpeg_hip_decoder_create(&dec, device);
		*out = dec;
		return err;
	}
	int encoder(int device, jpeg_hip_encoder **out)
	{
		const int err = enc ? RET_SUCCESS : jpeg_hip_encoder_create(&enc, device);
		*out = enc;
		return err;
	}
};

static StageObjects &stage_objects(int device)
{
	static std::mutex all_m;
	static std::map<int, StageObjects> all;
	std::lock_guard<std::mutex> lk(all_m);
	return all[device];
}

extern "C" int jpeg_hip_read_scan(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                                  const jpeg_hip_htable dc[4], const jpeg_hip_htable ac[4],
                                  const uint8_t *ecs, size_t n, int32_t *const coef[JPEG_HIP_MAX_COMP], size_t *consumed)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!frame || !scan || !dc || !ac || !ecs || !coef) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	if (consumed) {
		*consumed = 0;
	}
	JhHeader h;
	memset(&h, 0, sizeof h);
	err = geom_from_desc(frame, &h.g);
	if (err) {
		return err;
	}
	if (scan->Ns < 1 || scan->Ns > frame->Nf) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	ScanDesc sd;
	memset(&sd, 0, sizeof sd);
	unsigned seen = 0;
	for (int j = 0; j < JH_MAX_COMP; ++j) {
		sd.cs[j] = -1;
	}
	for (int j = 0; j < scan->Ns; ++j) {
		const int k = scan->Cs[j];
		if (k < 0 || k >= frame->Nf || (seen & (1u << k)) || scan->Td[j] < 0 || scan->Td[j] > 3 ||
		    scan->Ta[j] < 0 || scan->Ta[j] > 3) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		seen |= 1u << k;
		sd.cs[j] = k;
		sd.Td[k] = h.g.Td[k] = scan->Td[j];
		sd.Ta[k] = h.g.Ta[k] = scan->Ta[j];
	}
	sd.ncs = scan->Ns;
	if (scan->Ns == frame->Nf) {
		sd.comp = -1;                                      // every component: the MCU follows the scan header's order
		for (int j = 0; j < scan->Ns; ++j) {
			h.g.scan_order[j] = scan->Cs[j];
		}
	} else {
		sd.comp = scan->Ns == 1 ? scan->Cs[0] : -2;
	}
	for (int th = 0; th < 4; ++th) {
		spec_from_htable(&dc[th], &sd.ht[0][th]);
		spec_from_htable(&ac[th], &sd.ht[1][th]);
	}
	sd.ecs_begin = 0;
	sd.ecs_end = n;
	sd.Ri = scan->restart_interval > 0 ? scan->restart_interval : 0;
	h.g.Ri = sd.Ri;
	h.Ns = scan->Ns;
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			h.Q[t][i] = 1;                                 // not used: the run stops at the coefficients
		}
	}

	StageObjects &so = stage_objects(device);
	std::lock_guard<std::mutex> lease(so.m);
	jpeg_hip_decoder *dec = nullptr;
	err = so.decoder(device, &dec);
	if (err) {
		return err;
	}
	dec->coefs_only = true;
	const uint8_t *bufs[1] = {ecs};
	const size_t lens[1] = {n};
	err = decoder_prepare(dec, 1, bufs, lens, [&](int, JhHeader *ho, std::vector<ScanDesc> *sc) {
		*ho = h;
		sc->assign(1, sd);
		return RET_SUCCESS;
	});
	int status = RET_SUCCESS;
	if (!err) {
		dec->last_out = nullptr;
		dec->last_stride = 0;
		dec->last_stream = nullptr;
		dec->timer.reset();
		err = decoder_launch(dec, nullptr);
		dec->ran = err == RET_SUCCESS;
	}
	if (!err) {
		jpeg_hip_decoder_sync(dec, &status);
		if (consumed) {
			*consumed = dec->ecs_len[0];
		}
		// read_ecs() fails with what read_marker() returns at the end of the data; the blocks decoded until
		// then are in the buffers all the same
		for (int j = 0; j < scan->Ns && !err; ++j) {
			if (coef[scan->Cs[j]]) {
				err = jpeg_hip_decoder_get_coefficients(dec, 0, scan->Cs[j], coef[scan->Cs[j]]);
			}
		}
	}
	return err ? err : status;
}

extern "C" int jpeg_hip_decode_pixels(int device, const jpeg_hip_frame_desc *frame, const uint16_t Q[4][64],
                                      const int32_t *const coef[JPEG_HIP_MAX_COMP], void *samples)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!frame || !Q || !coef || !samples) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	JhGeom g;
	err = geom_from_desc(frame, &g);
	if (!err) {
		err = jh_finish_geometry(&g);
	}
	if (err) {
		return err;
	}
	const size_t ncoef = (size_t)g.nblk * 64;
	std::vector<int16_t> fc(ncoef, 0);
	for (int k = 0; k < g.ncomp; ++k) {
		if (!coef[k]) {
			return RET_FAILURE_LOGIC_ERROR;
		}
		if (!reference_layout_to_coefs(g, k, coef[k], fc.data())) {
			return RET_FAILURE_FILE_UNSUPPORTED;           // beyond 16 bits: no baseline stream holds such a coefficient
		}
	}
	JhQuant q;
	for (int t = 0; t < 4; ++t) {
		for (int z = 0; z < 64; ++z) {
			q.q[t][z] = Q[t][jh_zigzag[z]];
		}
	}
	const size_t body = (size_t)g.X * g.Y * g.out_comps * g.sample_bytes;
	StageObjects &so = stage_objects(device);
	std::lock_guard<std::mutex> lease(so.m);
	int16_t *d_coefs = (int16_t *)so.buffer(0, ncoef * sizeof(int16_t));
	JhQuant *d_quant = (JhQuant *)so.buffer(1, sizeof q);
	void *d_out = so.buffer(2, body ? body : 1);
	hipError_t e = d_coefs && d_quant && d_out ? hipSuccess : hipErrorOutOfMemory;
	e = e == hipSuccess ? hipMemcpy(d_coefs, fc.data(), ncoef * sizeof(int16_t), hipMemcpyHostToDevice) : e;
	e = e == hipSuccess ? hipMemcpy(d_quant, &q, sizeof q, hipMemcpyHostToDevice) : e;
	int r = RET_SUCCESS;
	if (e != hipSuccess) {
		r = RET_FAILURE_MEMORY_ALLOCATION;
	} else if (jh_launch_decode_pixels(g, 1, d_coefs, d_quant, d_out, body, nullptr)) {
		r = RET_FAILURE_LOGIC_ERROR;
	} else if (hipMemcpy(samples, d_out, body, hipMemcpyDeviceToHost) != hipSuccess) {
		r = RET_FAILURE_LOGIC_ERROR;
	}
	return r;
}

// frames the reference encoder sets up (encoder.c:109-146), and its parameters
static int encoder_frame_params(const jpeg_hip_frame_desc *fd, jpeg_hip_params *prm)
{
	if (!fd || (fd->Nf != 1 && fd->Nf != 3)) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	memset(prm, 0, sizeof *prm);
	prm->H = prm->V = 1;
	prm->q = 75;
	if (fd->Nf == 3) {
		if (fd->H[1] != 1 || fd->V[1] != 1 || fd->H[2] != 1 || fd->V[2] != 1 || fd->H[0] < 1 || fd->H[0] > 2 ||
		    fd->V[0] < 1 || fd->V[0] > 2 || fd->Tq[0] != 0 || fd->Tq[1] != 1 || fd->Tq[2] != 1) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
		prm->H = (uint8_t)fd->H[0];
		prm->V = (uint8_t)fd->V[0];
	} else if (fd->Tq[0] != 0) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_encode_pixels(int device, const jpeg_hip_frame_desc *frame, const uint16_t Q[4][64],
                                      const void *samples, int32_t *const coef[JPEG_HIP_MAX_COMP])
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!frame || !Q || !samples || !coef) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	jpeg_hip_params prm;
	err = encoder_frame_params(frame, &prm);
	if (err) {
		return err;
	}
	for (int t = 0; t < (frame->Nf > 1 ? 2 : 1); ++t) {
		for (int i = 0; i < 64; ++i) {
			if (Q[t][i] == 0) {
				return RET_FAILURE_FILE_UNSUPPORTED;       // quantize_block() would divide by zero
			}
		}
	}
	StageObjects &so = stage_objects(device);
	std::lock_guard<std::mutex> lease(so.m);
	jpeg_hip_encoder *enc = nullptr;
	err = so.encoder(device, &enc);
	if (err) {
		return err;
	}
	enc->restart_interval = 0;
	err = encoder_configure(enc, 1, frame->X, frame->Y, frame->Nf, frame->P, &prm, Q, nullptr);
	if (!err) {
		const JhGeom &g = enc->g;
		const size_t body = (size_t)g.X * g.Y * g.Nf * g.sample_bytes;
		void *d_in = so.buffer(0, body ? body : 1);
		JhEncState st0;
		memset(&st0, 0, sizeof st0);
		if (!d_in || hipMemcpy(d_in, samples, body, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemset(enc->d_state, 0, sizeof(JhEncState)) != hipSuccess) {
			err = RET_FAILURE_MEMORY_ALLOCATION;
		} else if (jh_launch_encode_pixels(g, 1, d_in, body, enc->d_quant, enc->d_coefs, nullptr, nullptr, nullptr, nullptr, enc->d_state, nullptr) ||
		           hipMemcpy(&st0, enc->d_state, sizeof st0, hipMemcpyDeviceToHost) != hipSuccess) {
			err = RET_FAILURE_LOGIC_ERROR;
		} else if (st0.status & JH_EST_NO_CODE) {
			err = RET_FAILURE_FILE_UNSUPPORTED;            // P > 12: a coefficient beyond 16 bits (see jpeg_amd.h)
		}
		enc->ran = !err && hipEventRecord(enc->done, nullptr) == hipSuccess;
		for (int k = 0; k < g.ncomp && !err; ++k) {
			if (coef[k]) {
				err = jpeg_hip_encoder_get_coefficients(enc, 0, k, coef[k]);
			}
		}
	}
	return err;
}

// what jpeg_hip_write_scan() and jpeg_hip_adapt_tables() share: an encoder object for one frame of the caller's
// description with the caller's coefficients uploaded (scan order, zig-zag, int16)
static int scan_encoder(StageObjects &so, int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                        const jpeg_hip_htable dc[4], const jpeg_hip_htable ac[4], int optimize,
                        const int32_t *const coef[JPEG_HIP_MAX_COMP], jpeg_hip_encoder **out_enc)
{
	*out_enc = nullptr;
	jpeg_hip_params prm;
	int err = encoder_frame_params(frame, &prm);
	if (err) {
		return err;
	}
	// the reference encoder's one scan: all components in frame order, tables 0 / 1 / 1 (encoder.c:139-141, 329-346)
	if (scan->Ns != frame->Nf || scan->restart_interval < 0 || scan->restart_interval > 65535) {
		return RET_FAILURE_FILE_UNSUPPORTED;
	}
	for (int j = 0; j < scan->Ns; ++j) {
		const int want = j == 0 ? 0 : 1;
		if (scan->Cs[j] != j || scan->Td[j] != want || scan->Ta[j] != want || !coef[j]) {
			return RET_FAILURE_FILE_UNSUPPORTED;
		}
	}
	prm.optimize = optimize != 0;
	JhHuffSpec ht[2][4];
	for (int th = 0; th < 4; ++th) {
		const int src = frame->Nf > 1 ? (th & 1) : 0;      // one component: table 0 only
		spec_from_htable(&dc[src], &ht[0][th]);
		spec_from_htable(&ac[src], &ht[1][th]);
	}
	uint16_t Q[4][64];
	for (int t = 0; t < 4; ++t) {
		for (int i = 0; i < 64; ++i) {
			Q[t][i] = 1;                                   // the headers built on the side are not used
		}
	}
	jpeg_hip_encoder *enc = nullptr;
	err = so.encoder(device, &enc);
	if (err) {
		return err;
	}
	enc->restart_interval = scan->restart_interval;
	err = encoder_configure(enc, 1, frame->X, frame->Y, frame->Nf, frame->P, &prm, Q, optimize ? nullptr : ht);
	if (!err) {
		const JhGeom &g = enc->g;
		const size_t ncoef = (size_t)g.nblk * 64;
		std::vector<int16_t> fc(ncoef, 0);
		bool fits = true;
		for (int k = 0; k < g.ncomp; ++k) {
			fits = reference_layout_to_coefs(g, k, coef[k], fc.data()) && fits;
		}
		if (!fits) {
			err = JPEG_HIP_ERR_NO_CODE;                    // category 16 and up: no table has a code (coeffs.c:56-74)
		} else if (hipMemcpy(enc->d_coefs, fc.data(), ncoef * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess ||
		           hipMemsetAsync(enc->d_state, 0, sizeof(JhEncState), nullptr) != hipSuccess) {
			err = RET_FAILURE_LOGIC_ERROR;
		}
	}
	if (err) {
		return err;
	}
	*out_enc = enc;
	return RET_SUCCESS;
}

// the tables the device derived (k_enc_adapt), for the caller's DHT segments
static int fetch_adapted_tables(jpeg_hip_encoder *enc, int Nf, jpeg_hip_htable dc[4], jpeg_hip_htable ac[4])
{
	JhDevSpec specs[4];
	if (hipMemcpy(specs, enc->d_specs, sizeof specs, hipMemcpyDeviceToHost) != hipSuccess) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	for (int th = 0; th < (Nf > 1 ? 2 : 1); ++th) {
		memcpy(dc[th].L, specs[0 * 2 + th].L, 16);
		memcpy(dc[th].V, specs[0 * 2 + th].V, 256);
		memcpy(ac[th].L, specs[1 * 2 + th].L, 16);
		memcpy(ac[th].V, specs[1 * 2 + th].V, 256);
	}
	return RET_SUCCESS;
}

extern "C" int jpeg_hip_write_scan(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                                   jpeg_hip_htable dc[4], jpeg_hip_htable ac[4], int optimize,
                                   const int32_t *const coef[JPEG_HIP_MAX_COMP], uint8_t *out, size_t cap, size_t *n)
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!frame || !scan || !dc || !ac || !coef || !n || (!out && cap)) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	*n = 0;
	jpeg_hip_encoder *enc = nullptr;
	StageObjects &so = stage_objects(device);
	std::lock_guard<std::mutex> lease(so.m);
	err = scan_encoder(so, device, frame, scan, dc, ac, optimize, coef, &enc);
	if (err) {
		return err;
	}
	const JhGeom &g = enc->g;
	const JhEncodeWork &w = enc->work;
	if ((optimize && (jh_launch_enc_histogram(w, g, nullptr, nullptr) || jh_launch_enc_adapt(w, g, nullptr, nullptr))) ||
	    jh_launch_enc_entropy(w, g, nullptr, nullptr, false)) {
		err = RET_FAILURE_LOGIC_ERROR;
	}
	if (!err) {
		enc->last_stream = nullptr;
		enc->stream_dst = nullptr;
		err = encoder_queue_results(enc, nullptr, nullptr);
	}
	if (!err) {
		enc->ran = true;
		int status = RET_SUCCESS;
		size_t total = 0;
		err = jpeg_hip_encoder_sync(enc, &status, &total);
		if (!err) {
			const size_t ecs = enc->state[0].ecs_bytes;
			*n = ecs;
			if (ecs > cap) {
				err = RET_FAILURE_OVERFLOW_ERROR;
			} else if (hipMemcpy(out, enc->d_out + enc->hdr_len[0], ecs, hipMemcpyDeviceToHost) != hipSuccess) {
				err = RET_FAILURE_LOGIC_ERROR;
			}
		}
		if (!err && optimize) {
			err = fetch_adapted_tables(enc, frame->Nf, dc, ac);
		}
	}
	return err;
}

extern "C" int jpeg_hip_adapt_tables(int device, const jpeg_hip_frame_desc *frame, const jpeg_hip_scan_desc *scan,
                                     jpeg_hip_htable dc[4], jpeg_hip_htable ac[4],
                                     const int32_t *const coef[JPEG_HIP_MAX_COMP])
{
	int err = require_device(device);
	if (err) {
		return err;
	}
	if (!frame || !scan || !dc || !ac || !coef) {
		return RET_FAILURE_LOGIC_ERROR;
	}
	jpeg_hip_encoder *enc = nullptr;
	StageObjects &so = stage_objects(device);
	std::lock_guard<std::mutex> lease(so.m);
	err = scan_encoder(so, device, frame, scan, dc, ac, 1, coef, &enc);
	if (err) {
		return err;
	}
	if (jh_launch_enc_histogram(enc->work, enc->g, nullptr, nullptr) || jh_launch_enc_adapt(enc->work, enc->g, nullptr, nullptr) ||
	    hipStreamSynchronize(nullptr) != hipSuccess) {
		err = RET_FAILURE_LOGIC_ERROR;
	}
	if (!err) {
		err = fetch_adapted_tables(enc, frame->Nf, dc, ac);
	}
	return err;
}

extern "C" void jpeg_hip_set_qtable(uint16_t Q[64], int chroma, int q)
{
	jh_set_qtable(Q, chroma != 0, q);
}

extern "C" void jpeg_hip_encoder_set_restart_interval(jpeg_hip_encoder *enc, int mcus)
{
	if (enc) {
		enc->restart_interval = mcus;
	}
}


static int slurp(FILE *f, std::vector<uint8_t> *out)
{
	uint8_t buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, f)) > 0) {
		out->insert(out->end(), buf, buf + n);
	}
	return ferror(f) ? RET_FAILURE_FILE_IO : RET_SUCCESS;
}

// decoder.c:661-685 process_jpeg_stream: one SOI..EOI codestream -> PPM/PGM at `path`
extern "C" int jpeg_hip_process_jpeg_stream(FILE *stream, const char *path)
{
	std::vector<uint8_t> data;
	int err = slurp(stream, &data);
	if (err) {
		return err;
	}
	jpeg_hip_decoder *dec = nullptr;
	err = jpeg_hip_decoder_create(&dec, 0);
	if (err) {
		fprintf(stderr, "malloc failure\n");
		return err;
	}
	const uint8_t *p = data.data();
	size_t n = data.size();
	void *d_out = nullptr;
	std::vector<uint8_t> body;
	jpeg_hip_frame_info fi;
	err = jpeg_hip_decoder_prepare(dec, 1, &p, &n);
	if (!err) {
		jpeg_hip_decoder_info(dec, &fi);
		const size_t fb = jpeg_hip_decoder_frame_bytes(dec);
		if (hipMalloc(&d_out, fb ? fb : 1) != hipSuccess) {
			err = RET_FAILURE_MEMORY_ALLOCATION;
		}
		if (!err) {
			err = jpeg_hip_decoder_run(dec, d_out, fb, nullptr);
		}
		int st = 0;
		if (!err) {
			err = jpeg_hip_decoder_sync(dec, &st);
		}
		if (!err) {
			body.resize(fb);
			if (hipMemcpy(body.data(), d_out, fb, hipMemcpyDeviceToHost) != hipSuccess) {
				err = RET_FAILURE_LOGIC_ERROR;
			}
		}
	}
	if (!err) {
		// write_frame (frame.c:548-567) + write_frame_header (:407-429)
		const bool gray = fi.out_components == 1;
		FILE *o = fopen(path ? path : (gray ? "output.pgm" : "output.ppm"), "w");
		if (!o) {
			err = RET_FAILURE_FILE_OPEN;
		} else {
			if (fprintf(o, "%s\n%u %u\n%i\n", gray ? "P5" : "P6", (unsigned)fi.X, (unsigned)fi.Y, (1 << fi.P) - 1) < 0 ||
			    fwrite(body.data(), 1, body.size(), o) < body.size()) {
				err = RET_FAILURE_FILE_IO;
			}
			fclose(o);
		}
	}
	if (d_out) {
		hipFree(d_out);
	}
	jpeg_hip_decoder_destroy(dec);
	return err;
}

// decoder.c:687-701
extern "C" int jpeg_hip_process_jpeg_file(const char *i_path, const char *o_path)
{
	FILE *stream = fopen(i_path, "r");
	if (stream == NULL) {
		fprintf(stderr, "fopen failure\n");
		return RET_FAILURE_FILE_OPEN;
	}
	int err = jpeg_hip_process_jpeg_stream(stream, o_path);
	fclose(stream);
	return err;
}

// frame.c:431-522 read_frame_header restated on a byte buffer
static int parse_pnm_header(const std::vector<uint8_t> &d, size_t *pos, int *comps, int *X, int *Y, int *P)
{
	size_t i = 0;
	auto skip_ws = [&]() { while (i < d.size() && isspace(d[i])) ++i; };
	auto skip_comments = [&]() {
		while (i < d.size() && d[i] == '#') {
			while (i < d.size() && d[i] != '\n') ++i;
			if (i < d.size()) ++i;
		}
	};
	auto number = [&](long *v) -> bool {
		skip_ws();
		if (i >= d.size() || !isdigit(d[i])) return false;
		long r = 0;
		while (i < d.size() && isdigit(d[i])) { r = r * 10 + (d[i] - '0'); if (r > 1000000) return false; ++i; }
		*v = r;
		return true;
	};
	if (d.size() < 2) return RET_FAILURE_FILE_IO;
	if (d[0] != 'P') return RET_FAILURE_FILE_UNSUPPORTED;
	if (d[1] == '5') *comps = 1; else if (d[1] == '6') *comps = 3; else return RET_FAILURE_FILE_UNSUPPORTED;
	i = 2;
	skip_ws();
	skip_comments();
	long w, h, maxval;
	if (!number(&w)) return RET_FAILURE_FILE_IO;
	skip_ws();
	skip_comments();
	if (!number(&h)) return RET_FAILURE_FILE_IO;
	skip_ws();
	skip_comments();
	if (!number(&maxval) || maxval <= 0) return RET_FAILURE_FILE_IO;
	int prec = 0;
	for (long m = maxval; m; m >>= 1) ++prec;              // floor_log2 + 1 (frame.c:259-275)
	if (prec > 16) return RET_FAILURE_FILE_UNSUPPORTED;
	skip_comments();
	if (i >= d.size() || !isspace(d[i])) return RET_FAILURE_FILE_UNSUPPORTED;
	++i;
	if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return RET_FAILURE_FILE_UNSUPPORTED;
	*X = (int)w; *Y = (int)h; *P = prec; *pos = i;
	return RET_SUCCESS;
}

// encoder.c:647-667 process_stream: PPM/PGM in, JPEG out
extern "C" int jpeg_hip_process_stream(FILE *i_stream, FILE *o_stream, const jpeg_hip_params *params)
{
	std::vector<uint8_t> data;
	int err = slurp(i_stream, &data);
	if (err) {
		return err;
	}
	size_t pos = 0;
	int comps, X, Y, P;
	err = parse_pnm_header(data, &pos, &comps, &X, &Y, &P);
	if (err) {
		return err;
	}
	const size_t ss = P <= 8 ? 1 : 2;
	const size_t body = (size_t)X * Y * comps * ss;
	if (data.size() - pos < body) {
		return RET_FAILURE_FILE_IO;                        // frame.c:296-299
	}
	jpeg_hip_encoder *enc = nullptr;
	err = jpeg_hip_encoder_create(&enc, 0);
	if (err) {
		return err;
	}
	void *d_in = nullptr;
	std::vector<uint8_t> jpg;
	err = jpeg_hip_encoder_configure(enc, 1, X, Y, comps, P, params);
	if (!err && hipMalloc(&d_in, body ? body : 1) != hipSuccess) {
		err = RET_FAILURE_MEMORY_ALLOCATION;
	}
	if (!err && hipMemcpy(d_in, data.data() + pos, body, hipMemcpyHostToDevice) != hipSuccess) {
		err = RET_FAILURE_LOGIC_ERROR;
	}
	if (!err) {
		err = jpeg_hip_encoder_run(enc, d_in, body, nullptr);
	}
	size_t len = 0;
	int st = 0;
	if (!err) {
		err = jpeg_hip_encoder_sync(enc, &st, &len);
	}
	if (!err) {
		jpg.resize(len);
		err = jpeg_hip_encoder_get_jpeg(enc, 0, jpg.data(), jpg.size(), &len);
	}
	if (!err && fwrite(jpg.data(), 1, len, o_stream) < len) {
		err = RET_FAILURE_FILE_IO;
	}
	if (d_in) {
		hipFree(d_in);
	}
	jpeg_hip_encoder_destroy(enc);
	return err;
}
