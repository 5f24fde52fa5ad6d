"""ctypes binding of libjpeg_amd.so (C-ABI declared in include/jpeg_amd.h).

PyTorch is used only as plumbing: device buffers (torch tensors on cuda:N) and the current HIP
stream.  The codec itself is the shared library; if it cannot be loaded, or no GPU is visible,
everything here raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JPEG_AMD_LIB") or os.path.join(HERE, "libjpeg_amd.so")   # the override is for experiments (tools/)

RET_SUCCESS = 0
RET_FAILURE_FILE_IO = 0x1000
RET_FAILURE_FILE_UNSUPPORTED = 0x1001
RET_FAILURE_MEMORY_ALLOCATION = 0x2000
RET_FAILURE_LOGIC_ERROR = 0x3000
RET_FAILURE_OVERFLOW_ERROR = 0x3001
RET_FAILURE_NO_MORE_DATA = 0x4000
JPEG_HIP_ERR_DC_RANGE = 0x5001
JPEG_HIP_ERR_NO_CODE = -1

MAX_COMP = 4
ABI_VERSION = 4          # JPEG_HIP_ABI_VERSION of include/jpeg_amd.h this file mirrors

EXPORTS = [
    "jpeg_hip_device_count", "jpeg_hip_version", "jpeg_hip_abi_version", "jpeg_hip_parse_info",
    "jpeg_hip_decoder_create", "jpeg_hip_decoder_destroy", "jpeg_hip_decoder_prepare",
    "jpeg_hip_decoder_prepare_pinned", "jpeg_hip_host_alloc", "jpeg_hip_host_free",
    "jpeg_hip_decoder_info", "jpeg_hip_decoder_frame_bytes", "jpeg_hip_decoder_run",
    "jpeg_hip_decoder_sync", "jpeg_hip_decoder_get_coefficients", "jpeg_hip_decoder_set_timing",
    "jpeg_hip_decoder_get_timing", "jpeg_hip_decoder_ecs_bytes", "jpeg_hip_decoder_get_stats",
    "jpeg_hip_decoder_set_debug", "jpeg_hip_decoder_get_chain_steps", "jpeg_hip_decoder_set_tolerant", "jpeg_hip_decoder_set_upload_stream", "jpeg_hip_decoder_set_lanes", "jpeg_hip_mjpeg_split",
    "jpeg_hip_encoder_create", "jpeg_hip_encoder_destroy", "jpeg_hip_encoder_configure",
    "jpeg_hip_encoder_info", "jpeg_hip_encoder_run", "jpeg_hip_encoder_sync",
    "jpeg_hip_encoder_run_stream", "jpeg_hip_encoder_sync_stream",
    "jpeg_hip_encoder_get_jpeg", "jpeg_hip_encoder_device_jpeg", "jpeg_hip_encoder_get_coefficients",
    "jpeg_hip_encoder_set_timing", "jpeg_hip_encoder_get_timing",
    "jpeg_hip_process_jpeg_stream", "jpeg_hip_process_jpeg_file", "jpeg_hip_process_stream",
    "jpeg_hip_read_scan", "jpeg_hip_decode_pixels", "jpeg_hip_encode_pixels", "jpeg_hip_write_scan",
    "jpeg_hip_adapt_tables", "jpeg_hip_set_qtable", "jpeg_hip_encoder_set_restart_interval",
    "jpeg_hip_encode_noalloc", "jpeg_hip_encode_noalloc_limit", "jpeg_hip_encode_noalloc_init",
    "jpeg_hip_selftest_divide",
    "jpeg_hip_selftest_ycc8",
    "jpeg_hip_streams_create", "jpeg_hip_streams_destroy",
    "jpeg_hip_decoder_set_float_tap", "jpeg_hip_encoder_set_float_tap",
    "jpeg_hip_dequantize_blocks", "jpeg_hip_quantize_blocks", "jpeg_hip_inverse_dct_blocks", "jpeg_hip_forward_dct_blocks",
    "jpeg_hip_frame_to_rgb", "jpeg_hip_frame_to_ycc",
]


class FrameInfo(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("X", C.c_int), ("Y", C.c_int), ("Nf", C.c_int),
        ("id", C.c_int * MAX_COMP), ("H", C.c_int * MAX_COMP), ("V", C.c_int * MAX_COMP),
        ("Tq", C.c_int * MAX_COMP), ("Td", C.c_int * MAX_COMP), ("Ta", C.c_int * MAX_COMP),
        ("max_H", C.c_int), ("max_V", C.c_int), ("m_x", C.c_int), ("m_y", C.c_int),
        ("size_x", C.c_int), ("size_y", C.c_int),
        ("bx", C.c_int * MAX_COMP), ("by", C.c_int * MAX_COMP),
        ("Ri", C.c_int), ("blocks_per_mcu", C.c_int), ("out_components", C.c_int), ("sample_bytes", C.c_int),
    ]


class Params(C.Structure):
    """struct params of the reference (encoder.c:68-76), same layout."""
    _fields_ = [("H", C.c_uint8), ("V", C.c_uint8), ("q", C.c_int), ("optimize", C.c_int)]


class FrameDesc(C.Structure):
    """jpeg_hip_frame_desc: what `struct context` knows about the frame (common.h:124-156)."""
    _fields_ = [("X", C.c_int), ("Y", C.c_int), ("P", C.c_int), ("Nf", C.c_int),
                ("H", C.c_int * MAX_COMP), ("V", C.c_int * MAX_COMP), ("Tq", C.c_int * MAX_COMP)]


class HTable(C.Structure):
    """jpeg_hip_htable: BITS and HUFFVAL of one DHT table (struct htable, common.h:85-91)."""
    _fields_ = [("L", C.c_uint8 * 16), ("V", C.c_uint8 * 256)]


class ScanDesc(C.Structure):
    """jpeg_hip_scan_desc (struct scan, decoder.c:186-195)."""
    _fields_ = [("Ns", C.c_int), ("Cs", C.c_int * MAX_COMP), ("Td", C.c_int * MAX_COMP), ("Ta", C.c_int * MAX_COMP),
                ("restart_interval", C.c_int)]


class JpegError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed: RET 0x%x" % (what, code & 0xffffffff))
        self.code = code


_lib = None


def lib():
    """Load the shared library (loudly)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `make` (or __graft_entry__.build()) first" % LIB_PATH)
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7.  Import torch
        # FIRST so that our library binds to that already-loaded runtime (same SONAME) and device
        # pointers / streams are shared; loading /opt/rocm's copy first leaves torch without a GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, sz, ip = C.c_void_p, C.c_size_t, C.POINTER(C.c_int)
        L.jpeg_hip_version.restype = C.c_char_p
        if L.jpeg_hip_abi_version() != ABI_VERSION:
            raise RuntimeError("%s has ABI %d, this binding was written for %d (include/jpeg_amd.h: JPEG_HIP_ABI_VERSION)"
                               % (LIB_PATH, L.jpeg_hip_abi_version(), ABI_VERSION))
        L.jpeg_hip_parse_info.argtypes = [C.c_char_p, sz, C.c_int, C.POINTER(FrameInfo)]
        L.jpeg_hip_decoder_create.argtypes = [C.POINTER(vp), C.c_int]
        L.jpeg_hip_decoder_destroy.argtypes = [vp]
        L.jpeg_hip_decoder_destroy.restype = None
        L.jpeg_hip_decoder_prepare.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(sz)]
        L.jpeg_hip_decoder_prepare_pinned.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]
        L.jpeg_hip_host_alloc.argtypes = [sz]
        L.jpeg_hip_host_alloc.restype = vp
        L.jpeg_hip_host_free.argtypes = [vp]
        L.jpeg_hip_host_free.restype = None
        L.jpeg_hip_decoder_info.argtypes = [vp, C.POINTER(FrameInfo)]
        L.jpeg_hip_decoder_frame_bytes.argtypes = [vp]
        L.jpeg_hip_decoder_frame_bytes.restype = sz
        L.jpeg_hip_decoder_run.argtypes = [vp, vp, sz, vp]
        L.jpeg_hip_decoder_sync.argtypes = [vp, ip]
        L.jpeg_hip_decoder_get_coefficients.argtypes = [vp, C.c_int, C.c_int, vp]
        L.jpeg_hip_decoder_set_timing.argtypes = [vp, C.c_int]
        L.jpeg_hip_decoder_set_timing.restype = None
        L.jpeg_hip_decoder_get_timing.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        L.jpeg_hip_decoder_ecs_bytes.argtypes = [vp, C.POINTER(sz)]
        L.jpeg_hip_decoder_get_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
        L.jpeg_hip_decoder_set_debug.argtypes = [vp, C.c_int]
        L.jpeg_hip_decoder_set_debug.restype = None
        L.jpeg_hip_decoder_set_upload_stream.argtypes = [vp, vp, C.c_int]
        L.jpeg_hip_decoder_set_upload_stream.restype = None
        L.jpeg_hip_decoder_set_lanes.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
        L.jpeg_hip_decoder_set_tolerant.argtypes = [vp, C.c_int]
        L.jpeg_hip_decoder_set_tolerant.restype = None
        L.jpeg_hip_mjpeg_split.argtypes = [C.c_char_p, sz, C.POINTER(sz), C.POINTER(sz), C.c_int]
        L.jpeg_hip_encoder_create.argtypes = [C.POINTER(vp), C.c_int]
        L.jpeg_hip_encoder_destroy.argtypes = [vp]
        L.jpeg_hip_encoder_destroy.restype = None
        L.jpeg_hip_encoder_configure.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Params)]
        L.jpeg_hip_encoder_info.argtypes = [vp, C.POINTER(FrameInfo)]
        L.jpeg_hip_encoder_run.argtypes = [vp, vp, sz, vp]
        L.jpeg_hip_encoder_sync.argtypes = [vp, ip, C.POINTER(sz)]
        L.jpeg_hip_encoder_run_stream.argtypes = [vp, vp, sz, vp, sz, vp]
        L.jpeg_hip_encoder_sync_stream.argtypes = [vp, ip, C.POINTER(sz)]
        L.jpeg_hip_encoder_get_jpeg.argtypes = [vp, C.c_int, vp, sz, C.POINTER(sz)]
        L.jpeg_hip_encoder_device_jpeg.argtypes = [vp, C.c_int, C.POINTER(sz)]
        L.jpeg_hip_encoder_device_jpeg.restype = vp
        L.jpeg_hip_encoder_get_coefficients.argtypes = [vp, C.c_int, C.c_int, vp]
        L.jpeg_hip_encoder_set_timing.argtypes = [vp, C.c_int]
        L.jpeg_hip_encoder_set_timing.restype = None
        L.jpeg_hip_encoder_get_timing.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        L.jpeg_hip_process_jpeg_file.argtypes = [C.c_char_p, C.c_char_p]
        fdp, sdp, htp, cpp = C.POINTER(FrameDesc), C.POINTER(ScanDesc), C.POINTER(HTable), C.POINTER(vp)
        L.jpeg_hip_read_scan.argtypes = [C.c_int, fdp, sdp, htp, htp, C.c_char_p, sz, cpp, C.POINTER(sz)]
        L.jpeg_hip_decode_pixels.argtypes = [C.c_int, fdp, vp, cpp, vp]
        L.jpeg_hip_encode_pixels.argtypes = [C.c_int, fdp, vp, vp, cpp]
        L.jpeg_hip_write_scan.argtypes = [C.c_int, fdp, sdp, htp, htp, C.c_int, cpp, vp, sz, C.POINTER(sz)]
        L.jpeg_hip_adapt_tables.argtypes = [C.c_int, fdp, sdp, htp, htp, cpp]
        L.jpeg_hip_encode_noalloc.argtypes = [C.c_int, C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, sz, vp, vp]
        L.jpeg_hip_encode_noalloc_limit.restype = sz
        L.jpeg_hip_selftest_divide.argtypes = [C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_ulonglong)]
        L.jpeg_hip_selftest_ycc8.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
        L.jpeg_hip_streams_create.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
        L.jpeg_hip_streams_destroy.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        L.jpeg_hip_streams_destroy.restype = None
        L.jpeg_hip_dequantize_blocks.argtypes = [C.c_int, sz, vp, vp, vp]
        L.jpeg_hip_quantize_blocks.argtypes = [C.c_int, sz, vp, vp, vp]
        L.jpeg_hip_inverse_dct_blocks.argtypes = [C.c_int, sz, vp, C.c_int]
        L.jpeg_hip_forward_dct_blocks.argtypes = [C.c_int, sz, vp, C.c_int]
        L.jpeg_hip_frame_to_rgb.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, sz, sz, C.c_int]
        L.jpeg_hip_frame_to_ycc.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, sz, sz, C.c_int]
        for fn in (L.jpeg_hip_decoder_set_float_tap, L.jpeg_hip_encoder_set_float_tap):
            fn.argtypes = [vp, vp]
            fn.restype = None
        L.jpeg_hip_set_qtable.argtypes = [vp, C.c_int, C.c_int]
        L.jpeg_hip_set_qtable.restype = None
        L.jpeg_hip_encoder_set_restart_interval.argtypes = [vp, C.c_int]
        L.jpeg_hip_encoder_set_restart_interval.restype = None
        _lib = L
    return _lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("jpeg_amd needs a HIP device (torch.cuda.is_available() is False); no CPU fallback exists")
    return torch


def _stream_ptr(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class StreamSet:
    """n streams measured to run on hardware queues of their own, and beside the caller's current stream
    (jpeg_hip_streams_create), as torch.cuda.ExternalStream objects.  `proven` = how many really are."""

    def __init__(self, device, n, avoid_current=True):
        torch = _torch()
        self.L, self.device, self.n = lib(), device, n
        self.raw = (C.c_void_p * n)()
        avoid = _stream_ptr(device) if avoid_current else C.c_void_p()
        self.proven = self.L.jpeg_hip_streams_create(device, n, avoid, int(avoid_current), self.raw)
        if self.proven < 0:
            raise JpegError(RET_FAILURE_LOGIC_ERROR, "jpeg_hip_streams_create")
        self.streams = [torch.cuda.ExternalStream(int(self.raw[i]), device=device) for i in range(n)]

    def close(self):
        if self.raw is not None:
            self.L.jpeg_hip_streams_destroy(self.device, self.n, self.raw)
            self.raw, self.streams = None, []


def _timing(fn, handle):
    names = (C.c_char_p * 32)()
    ms = (C.c_float * 32)()
    n = fn(handle, names, ms, 32)
    return {names[i].decode(): float(ms[i]) for i in range(n)}


def parse_info(jpg, tolerant=False):
    """Headers and geometry of one codestream (host only) -> (RET code, FrameInfo)."""
    fi = FrameInfo()
    err = lib().jpeg_hip_parse_info(jpg, len(jpg), int(tolerant), C.byref(fi))
    return err, fi


class PinnedBuffer:
    """Page-locked host memory from the library (jpeg_hip_host_alloc): what a capture or I/O layer reads a Motion
    JPEG file into, so that its frames go to the device without a staging copy."""

    def __init__(self, nbytes):
        self.L = lib()
        self.nbytes = nbytes
        self.addr = self.L.jpeg_hip_host_alloc(nbytes)
        if not self.addr:
            raise JpegError(RET_FAILURE_MEMORY_ALLOCATION, "jpeg_hip_host_alloc")

    def view(self):
        return (C.c_ubyte * self.nbytes).from_address(self.addr)

    def close(self):
        if self.addr:
            self.L.jpeg_hip_host_free(self.addr)
            self.addr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedFrames:
    """Frames (offset, length) inside a PinnedBuffer; slicing gives the frames of a batch or of a rank's shard."""

    def __init__(self, buf, offsets, lengths):
        self.buf, self.offsets, self.lengths = buf, list(offsets), list(lengths)

    @classmethod
    def from_frames(cls, frames):
        """Copy codestreams (bytes) back to back into a new pinned buffer, as they lie in a Motion JPEG file."""
        lengths = [len(f) for f in frames]
        buf = PinnedBuffer(max(1, sum(lengths)))
        dst = np.frombuffer(buf.view(), dtype=np.uint8)
        offsets, at = [], 0
        for f in frames:
            dst[at:at + len(f)] = np.frombuffer(f, dtype=np.uint8)
            offsets.append(at)
            at += len(f)
        return cls(buf, offsets, lengths)

    def __len__(self):
        return len(self.offsets)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return PinnedFrames(self.buf, self.offsets[i], self.lengths[i])
        return C.string_at(self.buf.addr + self.offsets[i], self.lengths[i])

    def nbytes(self):
        return sum(self.lengths)

    def c_arrays(self):
        n = len(self.offsets)
        ptrs = (C.c_void_p * n)(*[self.buf.addr + o for o in self.offsets])
        lens = (C.c_size_t * n)(*self.lengths)
        return ptrs, lens


class Decoder:
    """Batch decoder: JPEG codestreams (same geometry) -> PPM/PGM bodies resident in HBM.

    Mirrors read_ecs()+epilogue() of the reference decoder (decoder.c:364-388, 456-470)."""

    def __init__(self, device=0):
        self.L = lib()
        self.device = device
        self.h = C.c_void_p()
        err = self.L.jpeg_hip_decoder_create(C.byref(self.h), device)
        if err:
            raise JpegError(err, "jpeg_hip_decoder_create")
        self.n = 0
        self.info = None

    def close(self):
        if self.h:
            self.L.jpeg_hip_decoder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, jpgs):
        """jpgs: codestreams as bytes-like objects, or a PinnedFrames (frames in page-locked memory: no staging copy)"""
        if isinstance(jpgs, PinnedFrames):
            self._keep = jpgs
            n = len(jpgs)
            ptrs, lens = jpgs.c_arrays()
            err = self.L.jpeg_hip_decoder_prepare_pinned(self.h, n, ptrs, lens)
        else:
            jpgs = [bytes(j) for j in jpgs]
            self._keep = jpgs
            n = len(jpgs)
            ptrs = (C.c_char_p * n)(*jpgs)
            lens = (C.c_size_t * n)(*[len(j) for j in jpgs])
            err = self.L.jpeg_hip_decoder_prepare(self.h, n, ptrs, lens)
        if err:
            raise JpegError(err, "jpeg_hip_decoder_prepare")
        self.n = n
        self.info = FrameInfo()
        self.L.jpeg_hip_decoder_info(self.h, C.byref(self.info))
        self.frame_bytes = self.L.jpeg_hip_decoder_frame_bytes(self.h)
        return self.info

    def set_timing(self, on=True):
        self.L.jpeg_hip_decoder_set_timing(self.h, int(on))

    def timing(self):
        return _timing(self.L.jpeg_hip_decoder_get_timing, self.h)

    def alloc_output(self):
        torch = _torch()
        return torch.empty((self.n, self.frame_bytes), dtype=torch.uint8, device="cuda:%d" % self.device)

    def run(self, out=None):
        """Launch the device pipeline on torch's current stream; returns the output tensor."""
        if out is None:
            out = self.alloc_output()
        err = self.L.jpeg_hip_decoder_run(self.h, C.c_void_p(out.data_ptr()), out.stride(0), _stream_ptr(self.device))
        if err:
            raise JpegError(err, "jpeg_hip_decoder_run")
        return out

    def set_float_tap(self, on=True):
        """Stage tap: the next run() also stores the samples as floats before rounding (frame->data behind
        frame_to_rgb, frame.c:188-244) -> float32 tensor [n, Y, X, 1 or 3] (returned), None = off."""
        torch = _torch()
        self.tap = None
        if on:
            i = self.info
            self.tap = torch.zeros((self.n, i.Y, i.X, i.out_components), dtype=torch.float32, device="cuda:%d" % self.device)
        self.L.jpeg_hip_decoder_set_float_tap(self.h, C.c_void_p(self.tap.data_ptr() if on else None))
        return self.tap

    def sync(self):
        st = (C.c_int * self.n)()
        err = self.L.jpeg_hip_decoder_sync(self.h, st)
        if err and not any(st):
            raise JpegError(err, "jpeg_hip_decoder_sync")  # failed before any frame status was written
        return list(st)

    def stats(self, frame=0):
        """(serial fallback used, workgroups repaired, blocks decoded, restart segments) of one frame"""
        out = (C.c_uint32 * 4)()
        err = self.L.jpeg_hip_decoder_get_stats(self.h, frame, out)
        if err:
            raise JpegError(err, "jpeg_hip_decoder_get_stats")
        return tuple(int(v) for v in out)

    def chain_steps(self, frame=0, cap=18):
        """diagnostics: chains by the number of further sub-sequences they walked"""
        h = (C.c_uint32 * cap)()
        err = self.L.jpeg_hip_decoder_get_chain_steps(self.h, frame, h, cap)
        if err:
            raise JpegError(err, "jpeg_hip_decoder_get_chain_steps")
        return list(h)

    def set_debug(self, flags):
        self.L.jpeg_hip_decoder_set_debug(self.h, int(flags))

    def set_upload_stream(self, stream):
        """prepare() uploads on this torch stream (the one run() will be called under) instead of the decoder's own
        copy stream; None: back to the copy stream."""
        if stream is None:
            self.L.jpeg_hip_decoder_set_upload_stream(self.h, None, 0)
        else:
            self.L.jpeg_hip_decoder_set_upload_stream(self.h, C.c_void_p(stream.cuda_stream), 1)
        self._upload_stream = stream                     # keeps it alive

    def set_lanes(self, n, streams=None):
        """run() splits the batch over n streams (its own, the current torch stream, and n - 1 of the decoder's that
        are measured to run beside it, or the n - 1 torch streams given): narrow kernels of one part beside the wide
        ones of the others.  Returns how many lanes really run side by side."""
        arr = None
        if streams is not None:
            assert len(streams) == n - 1
            arr = (C.c_void_p * max(1, n - 1))(*[C.c_void_p(s.cuda_stream) for s in streams])
        self._lane_streams = streams                     # kept alive
        got = self.L.jpeg_hip_decoder_set_lanes(self.h, int(n), _stream_ptr(self.device), arr)
        if got < 0:
            raise JpegError(RET_FAILURE_LOGIC_ERROR, "jpeg_hip_decoder_set_lanes")
        return got

    def set_tolerant(self, on=True):
        """Skip APP9/APP10/APP15/DNL/JPGn segments instead of failing like the reference (call before prepare)."""
        self.L.jpeg_hip_decoder_set_tolerant(self.h, int(on))

    def ecs_bytes(self):
        v = (C.c_size_t * self.n)()
        self.L.jpeg_hip_decoder_ecs_bytes(self.h, v)
        return list(v)

    def coefficients(self, frame):
        """Per component int32 [by*bx, 64] in the reference's int_buffer layout."""
        out = []
        for k in range(self.info.Nf):
            a = np.zeros((self.info.by[k] * self.info.bx[k], 64), np.int32)
            err = self.L.jpeg_hip_decoder_get_coefficients(self.h, frame, k, a.ctypes.data_as(C.c_void_p))
            if err:
                raise JpegError(err, "jpeg_hip_decoder_get_coefficients")
            out.append(a)
        return out

    def to_arrays(self, out):
        """Output tensor -> list of numpy sample arrays shaped like the PPM/PGM."""
        i = self.info
        host = out.cpu().numpy()
        res = []
        for f in range(self.n):
            raw = host[f].tobytes()
            dt = np.uint8 if i.sample_bytes == 1 else np.dtype(">u2")
            a = np.frombuffer(raw, dtype=dt).reshape(i.Y, i.X, i.out_components)
            a = a.astype(np.uint8 if i.sample_bytes == 1 else np.uint16)
            res.append(a[:, :, 0] if i.out_components == 1 else a)
        return res


class Encoder:
    """Batch encoder: PPM/PGM bodies resident in HBM -> complete JPEG codestreams in HBM.

    Mirrors prologue()+produce_codestream() of the reference encoder (encoder.c:176-193, 589-645)."""

    def __init__(self, device=0):
        self.L = lib()
        self.device = device
        self.h = C.c_void_p()
        err = self.L.jpeg_hip_encoder_create(C.byref(self.h), device)
        if err:
            raise JpegError(err, "jpeg_hip_encoder_create")
        self.n = 0

    def close(self):
        if self.h:
            self.L.jpeg_hip_encoder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, nframes, X, Y, Nf, P=8, H=2, V=1, q=75, optimize=1, restart_interval=0):
        prm = Params(H, V, q, optimize)
        self.L.jpeg_hip_encoder_set_restart_interval(self.h, int(restart_interval))
        err = self.L.jpeg_hip_encoder_configure(self.h, nframes, X, Y, Nf, P, C.byref(prm))
        if err:
            raise JpegError(err, "jpeg_hip_encoder_configure")
        self.n = nframes
        self.info = FrameInfo()
        self.L.jpeg_hip_encoder_info(self.h, C.byref(self.info))
        self.frame_bytes = X * Y * Nf * (1 if P <= 8 else 2)
        return self.info

    def set_timing(self, on=True):
        self.L.jpeg_hip_encoder_set_timing(self.h, int(on))

    def timing(self):
        return _timing(self.L.jpeg_hip_encoder_get_timing, self.h)

    def upload(self, images):
        """List of (Y,X[,3]) integer arrays -> device tensor of PPM/PGM bodies."""
        torch = _torch()
        P = self.info.P
        bodies = [np.asarray(a).astype(np.uint8).tobytes() if P <= 8 else np.asarray(a).astype(">u2").tobytes()
                  for a in images]
        host = np.frombuffer(b"".join(bodies), np.uint8).reshape(len(images), self.frame_bytes)
        return torch.from_numpy(host.copy()).to("cuda:%d" % self.device)

    def run(self, samples):
        err = self.L.jpeg_hip_encoder_run(self.h, C.c_void_p(samples.data_ptr()), samples.stride(0),
                                          _stream_ptr(self.device))
        if err:
            raise JpegError(err, "jpeg_hip_encoder_run")

    def sync(self):
        st = (C.c_int * self.n)()
        ln = (C.c_size_t * self.n)()
        err = self.L.jpeg_hip_encoder_sync(self.h, st, ln)
        if err and not any(st):
            raise JpegError(err, "jpeg_hip_encoder_sync")  # failed before any frame status was written
        return list(st), list(ln)

    def set_float_tap(self, on=True):
        """Stage tap: the next run() also stores the component planes behind the down-sampling, before the level
        shift (component[].frame_buffer, frame.c:84-132) -> list of float32 views [n, by*8, bx*8] per component."""
        torch = _torch()
        self.tap = None
        if not on:
            self.L.jpeg_hip_encoder_set_float_tap(self.h, None)
            return None
        i = self.info
        sizes = [i.by[k] * 8 * i.bx[k] * 8 for k in range(i.Nf)]
        self.tap = torch.zeros((self.n, sum(sizes)), dtype=torch.float32, device="cuda:%d" % self.device)
        self.L.jpeg_hip_encoder_set_float_tap(self.h, C.c_void_p(self.tap.data_ptr()))
        views, at = [], 0
        for k in range(i.Nf):
            views.append(self.tap[:, at:at + sizes[k]].reshape(self.n, i.by[k] * 8, i.bx[k] * 8))
            at += sizes[k]
        return views

    def run_stream(self, samples, out, cap=None):
        """run(), and the finished codestreams packed back to back into `out`: a PinnedBuffer (page-locked host
        memory: the kernels store across the link) or a uint8 device tensor (jpeg_hip_encoder_run_stream)."""
        if isinstance(out, PinnedBuffer):
            addr, room = out.addr, out.nbytes
        else:
            addr, room = out.data_ptr(), out.numel()
        cap = room if cap is None else min(cap, room)
        err = self.L.jpeg_hip_encoder_run_stream(self.h, C.c_void_p(samples.data_ptr()), samples.stride(0),
                                                 C.c_void_p(addr), cap, _stream_ptr(self.device))
        if err:
            raise JpegError(err, "jpeg_hip_encoder_run_stream")

    def sync_stream(self):
        """-> (per-frame status, frame_off[n + 1]): frame f is out[frame_off[f]:frame_off[f + 1]]."""
        st = (C.c_int * self.n)()
        off = (C.c_size_t * (self.n + 1))()
        err = self.L.jpeg_hip_encoder_sync_stream(self.h, st, off)
        if err and not any(st):
            raise JpegError(err, "jpeg_hip_encoder_sync_stream")
        return list(st), list(off)

    def jpeg(self, frame, length):
        buf = C.create_string_buffer(length)
        n = C.c_size_t()
        err = self.L.jpeg_hip_encoder_get_jpeg(self.h, frame, buf, length, C.byref(n))
        if err:
            raise JpegError(err, "jpeg_hip_encoder_get_jpeg")
        return buf.raw[:n.value]

    def coefficients(self, frame):
        out = []
        for k in range(self.info.Nf):
            a = np.zeros((self.info.by[k] * self.info.bx[k], 64), np.int32)
            err = self.L.jpeg_hip_encoder_get_coefficients(self.h, frame, k, a.ctypes.data_as(C.c_void_p))
            if err:
                raise JpegError(err, "jpeg_hip_encoder_get_coefficients")
            out.append(a)
        return out


def decode(jpgs, device=0):
    """Convenience: list of codestreams -> (list of sample arrays, list of RET codes)."""
    d = Decoder(device)
    try:
        d.prepare(jpgs)
        out = d.run()
        st = d.sync()
        return d.to_arrays(out), st
    finally:
        d.close()


def decode_mixed(jpgs, device=0):
    """Codestreams of ANY mix of sizes, sampling and scan layouts -> (list of sample arrays or None, list of RET
    codes), in the order given.  A decoder batch wants frames of one geometry whose scans agree in number and kind
    (jpeg_amd.h): the files are grouped by what their headers say (host only), each group is one batch, and a frame
    that a batch of several refuses as unsupported -- its scans differ from its neighbours' -- is decoded once more in
    a batch of its own.  The reference takes one file per call (decoder.c:661-685) and has no such notion."""
    groups, order = {}, []
    res, sts = [None] * len(jpgs), [RET_FAILURE_FILE_UNSUPPORTED] * len(jpgs)
    for i, j in enumerate(jpgs):
        err, fi = parse_info(j)
        if err:
            sts[i] = err
            continue
        key = (fi.P, fi.X, fi.Y, fi.Nf) + tuple((fi.id[k], fi.H[k], fi.V[k]) for k in range(fi.Nf))
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(i)
    d = Decoder(device)
    try:
        def batch(idx):
            try:
                d.prepare([jpgs[i] for i in idx])
            except JpegError as e:
                return [None] * len(idx), [e.code] * len(idx)
            out = d.run()
            st = d.sync()
            px = d.to_arrays(out)
            return [px[k] if st[k] == 0 else None for k in range(len(idx))], st
        for key in order:
            idx = groups[key]
            px, st = batch(idx)
            for k, i in enumerate(idx):
                res[i], sts[i] = px[k], st[k]
            if len(idx) > 1:
                for k, i in enumerate(idx):
                    if st[k] == RET_FAILURE_FILE_UNSUPPORTED:
                        p1, s1 = batch([i])
                        res[i], sts[i] = p1[0], s1[0]
        return res, sts
    finally:
        d.close()


def encode(images, P=8, H=2, V=1, q=75, optimize=1, device=0, restart_interval=0):
    """Convenience: list of same-shaped images -> (list of JPEG bytes, list of RET codes)."""
    a0 = np.asarray(images[0])
    Y, X = a0.shape[:2]
    Nf = 1 if a0.ndim == 2 else a0.shape[2]
    e = Encoder(device)
    try:
        e.configure(len(images), X, Y, Nf, P, H, V, q, optimize, restart_interval)
        e.run(e.upload(images))
        st, ln = e.sync()
        return [e.jpeg(f, ln[f]) if st[f] == 0 else None for f in range(len(images))], st
    finally:
        e.close()


# ---- stage level (jpeg_hip_read_scan / decode_pixels / encode_pixels / write_scan) -------------------------
# Thin ctypes wrappers on host numpy arrays: what a maintainer calls from inside the reference's own
# parse_format() / produce_codestream(), and what the stage-parity tests call.

def frame_desc(X, Y, P, H, V, Tq):
    fd = FrameDesc()
    fd.X, fd.Y, fd.P, fd.Nf = X, Y, P, len(H)
    for k in range(len(H)):
        fd.H[k], fd.V[k], fd.Tq[k] = H[k], V[k], Tq[k]
    return fd


def scan_desc(Cs, Td, Ta, restart_interval=0):
    sd = ScanDesc()
    sd.Ns = len(Cs)
    for j in range(len(Cs)):
        sd.Cs[j], sd.Td[j], sd.Ta[j] = Cs[j], Td[j], Ta[j]
    sd.restart_interval = restart_interval
    return sd


def htables(tabs):
    """tabs: up to four (L[16], V[...]) pairs or None -> HTable[4]."""
    arr = (HTable * 4)()
    for i, t in enumerate(tabs):
        if t is None:
            continue
        Lc, Vs = t
        for j in range(16):
            arr[i].L[j] = int(Lc[j])
        for j, v in enumerate(Vs):
            arr[i].V[j] = int(v)
    return arr


def _blocks(fd, k):
    """(b_y, b_x) of component k (common.c:171-182; one-component frames: the padded grid of that component)."""
    Nf = fd.Nf
    mh = max(fd.H[i] for i in range(Nf))
    mv = max(fd.V[i] for i in range(Nf))
    m_x = (fd.X + 8 * mh - 1) // (8 * mh)
    m_y = (fd.Y + 8 * mv - 1) // (8 * mv)
    return m_y * fd.V[k], m_x * fd.H[k]


def _coef_ptrs(arrs):
    ptrs = (C.c_void_p * MAX_COMP)()
    for k, a in enumerate(arrs):
        ptrs[k] = a.ctypes.data if a is not None else None
    return ptrs


def read_scan(fd, sd, dc, ac, ecs, device=0):
    """-> (RET code, {component: int32 [b_y*b_x, 64]}, offset of the marker that ended the scan)."""
    coefs = [None] * MAX_COMP
    for j in range(sd.Ns):
        by, bx = _blocks(fd, sd.Cs[j])
        coefs[sd.Cs[j]] = np.zeros((by * bx, 64), np.int32)
    used = C.c_size_t(0)
    err = lib().jpeg_hip_read_scan(device, C.byref(fd), C.byref(sd), dc, ac, ecs, len(ecs), _coef_ptrs(coefs), C.byref(used))
    return err, {sd.Cs[j]: coefs[sd.Cs[j]] for j in range(sd.Ns)}, used.value


def decode_pixels(fd, Q, coefs, device=0):
    """Q: uint16 [4, 64] raster order; coefs: list of int32 arrays per component -> (RET code, samples)."""
    Q = np.ascontiguousarray(Q, np.uint16)
    coefs = [np.ascontiguousarray(c, np.int32) for c in coefs]
    nc = 1 if fd.Nf == 1 else 3
    shape = (fd.Y, fd.X) if nc == 1 else (fd.Y, fd.X, 3)
    out = np.zeros(shape, np.uint8 if fd.P <= 8 else ">u2")
    err = lib().jpeg_hip_decode_pixels(device, C.byref(fd), Q.ctypes.data, _coef_ptrs(coefs), out.ctypes.data)
    return err, out.astype(np.uint8 if fd.P <= 8 else np.uint16)


def encode_pixels(fd, Q, samples, device=0):
    """samples: [Y, X] or [Y, X, 3] -> (RET code, list of int32 [b_y*b_x, 64] per component)."""
    Q = np.ascontiguousarray(Q, np.uint16)
    body = np.ascontiguousarray(samples, np.uint8 if fd.P <= 8 else ">u2")
    coefs = []
    for k in range(fd.Nf):
        by, bx = _blocks(fd, k)
        coefs.append(np.zeros((by * bx, 64), np.int32))
    err = lib().jpeg_hip_encode_pixels(device, C.byref(fd), Q.ctypes.data, body.ctypes.data, _coef_ptrs(coefs))
    return err, coefs


def write_scan(fd, sd, dc, ac, optimize, coefs, device=0):
    """-> (RET code, entropy-coded bytes); with optimize the tables come back in dc / ac."""
    coefs = [np.ascontiguousarray(c, np.int32) for c in coefs]
    cap = 4096 + sum(c.size for c in coefs) * 4
    out = np.zeros(cap, np.uint8)
    n = C.c_size_t(0)
    err = lib().jpeg_hip_write_scan(device, C.byref(fd), C.byref(sd), dc, ac, int(optimize), _coef_ptrs(coefs),
                                    out.ctypes.data, cap, C.byref(n))
    return err, out[:n.value].tobytes() if err == 0 else b""
