"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two checkers live behind this module:

* ``Oracle``    -- oracle/liboracle.so, our C restatement (oracle/jpeg_oracle.c);
* ``Reference`` -- oracle/_ref/libref_{dec,enc}.so, the REAL xbarin02/jpeg sources
  compiled in place by oracle/Makefile plus the tap harness (ref_tap_*.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (jpeg_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile liboracle.so (always possible) and oracle/_ref (only where /root/reference exists)."""
    lib = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(lib) or \
            os.path.getmtime(lib) < os.path.getmtime(os.path.join(HERE, "jpeg_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return all(os.path.exists(os.path.join(HERE, "_ref", f)) for f in ("libref_dec.so", "libref_enc.so"))


def pnm_bytes(samples, P):
    """PPM/PGM file bytes for an (Y, X) or (Y, X, 3) integer array (big-endian u16 when P > 8)."""
    a = np.asarray(samples)
    Y, X = a.shape[:2]
    magic = b"P5" if a.ndim == 2 or a.shape[2] == 1 else b"P6"
    maxval = (1 << P) - 1
    body = a.astype(np.uint8).tobytes() if maxval <= 255 else a.astype(">u2").tobytes()
    return magic + b"\n%d %d\n%d\n" % (X, Y, maxval) + body


def pnm_body(samples, P):
    a = np.asarray(samples)
    return a.astype(np.uint8).tobytes() if P <= 8 else a.astype(">u2").tobytes()


def parse_pnm(data):
    """Minimal P5/P6 reader -> (array, P)."""
    parts = data.split(b"\n", 3)
    magic, dims, maxval, body = parts[0], parts[1], int(parts[2]), parts[3]
    X, Y = (int(v) for v in dims.split())
    nc = 1 if magic == b"P5" else 3
    P = maxval.bit_length()
    dt = np.uint8 if maxval <= 255 else np.dtype(">u2")
    a = np.frombuffer(body, dtype=dt, count=X * Y * nc).reshape(Y, X, nc)
    return (a[:, :, 0] if nc == 1 else a).astype(np.uint16 if maxval > 255 else np.uint8), P


class _Image:
    """Coefficients and geometry captured from a codec run (oracle or reference)."""

    def __init__(self):
        self.P = self.X = self.Y = self.Nf = 0
        self.comps = []      # dicts: id,H,V,Tq,bx,by,coef (int32 [by*bx,64] raster order)
        self.qtables = None  # uint16 [4,64] raster order
        self.frame = None    # float32 [size_y,size_x,Nf] before rounding (decode only)
        self.planes = None   # encode only: float32 [by*8, bx*8] per component behind the down-sampling, before the level shift
        self.htables = None
        self.mblocks = 0


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        self.L = L
        L.orc_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(_u8p), C.POINTER(C.c_size_t)]
        L.orc_encode.argtypes = [C.c_char_p] + [C.c_int] * 9 + [C.POINTER(C.c_void_p), C.POINTER(_u8p), C.POINTER(C.c_size_t)]
        L.orc_encode_scans.argtypes = [C.c_char_p] + [C.c_int] * 11 + [C.POINTER(_u8p), C.POINTER(C.c_size_t)]
        L.orc_image_free.argtypes = [C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_img_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5 + [C.POINTER(C.c_size_t)] * 5
        L.orc_img_comp.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 6 + [C.POINTER(C.c_size_t)] * 2
        L.orc_img_coef.argtypes = [C.c_void_p, C.c_int]
        L.orc_img_coef.restype = C.POINTER(C.c_int32)
        L.orc_img_qtable.argtypes = [C.c_void_p, C.c_int]
        L.orc_img_qtable.restype = C.POINTER(C.c_uint16)
        L.orc_img_frame.argtypes = [C.c_void_p]
        L.orc_img_frame.restype = C.POINTER(C.c_float)
        L.orc_img_plane.argtypes = [C.c_void_p, C.c_int]
        L.orc_img_plane.restype = C.POINTER(C.c_float)
        L.orc_img_htable.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
        L.orc_decode_coeff.restype = C.c_int32
        L.orc_decode_coeff.argtypes = [C.c_int, C.c_int]
        L.orc_encode_cat.argtypes = [C.c_int32]
        L.orc_encode_extra.argtypes = [C.c_int32, C.c_int]

    # ---- whole codec -------------------------------------------------------------
    def _image(self, h):
        L = self.L
        img = _Image()
        iv = [C.c_int() for _ in range(5)]
        sv = [C.c_size_t() for _ in range(5)]
        L.orc_img_info(h, *[C.byref(v) for v in iv], *[C.byref(v) for v in sv])
        img.P, img.X, img.Y, img.Nf, ncomp = (v.value for v in iv)
        img.m_x, img.m_y, img.size_x, img.size_y, img.mblocks = (v.value for v in sv)
        for k in range(ncomp):
            ci = [C.c_int() for _ in range(6)]
            cs = [C.c_size_t() for _ in range(2)]
            L.orc_img_comp(h, k, *[C.byref(v) for v in ci], *[C.byref(v) for v in cs])
            bx, by = cs[0].value, cs[1].value
            coef = np.ctypeslib.as_array(L.orc_img_coef(h, k), shape=(bx * by, 64)).copy()
            img.comps.append(dict(id=ci[0].value, H=ci[1].value, V=ci[2].value, Tq=ci[3].value,
                                  Td=ci[4].value, Ta=ci[5].value, bx=bx, by=by, coef=coef))
        img.qtables = np.stack([np.ctypeslib.as_array(L.orc_img_qtable(h, t), shape=(64,)).copy() for t in range(4)])
        fp = L.orc_img_frame(h)
        if fp:
            img.frame = np.ctypeslib.as_array(fp, shape=(img.size_y, img.size_x, img.Nf)).copy()
        pl = [L.orc_img_plane(h, k) for k in range(ncomp)]
        if all(pl):
            img.planes = [np.ctypeslib.as_array(pl[k], shape=(img.comps[k]["by"] * 8, img.comps[k]["bx"] * 8)).copy()
                          for k in range(ncomp)]
        img.htables = {}
        for tc in range(2):
            for th in range(4):
                Lb = C.create_string_buffer(16)
                Vb = C.create_string_buffer(256)
                L.orc_img_htable(h, tc, th, Lb, Vb)
                img.htables[(tc, th)] = (np.frombuffer(Lb.raw, np.uint8).copy(), np.frombuffer(Vb.raw, np.uint8).copy())
        return img

    def decode(self, jpg):
        """-> (err, samples array or None, image info or None)."""
        h = C.c_void_p()
        out = _u8p()
        n = C.c_size_t()
        err = self.L.orc_decode(jpg, len(jpg), C.byref(h), C.byref(out), C.byref(n))
        if err:
            return err, None, None
        img = self._image(h)
        raw = C.string_at(out, n.value)
        self.L.orc_free(out)
        self.L.orc_image_free(h)
        nc = 1 if img.Nf == 1 else 3
        dt = np.uint8 if img.P <= 8 else np.dtype(">u2")
        a = np.frombuffer(raw, dtype=dt).reshape(img.Y, img.X, nc)
        a = a.astype(np.uint8 if img.P <= 8 else np.uint16)
        return 0, (a[:, :, 0] if nc == 1 else a), img

    def encode(self, samples, P=8, H=2, V=1, q=75, optimize=1, restart_interval=0):
        """samples: (Y,X) gray or (Y,X,3) RGB ints -> (err, jpeg bytes or None, image info)."""
        a = np.asarray(samples)
        Y, X = a.shape[:2]
        Nf = 1 if a.ndim == 2 else a.shape[2]
        body = pnm_body(a, P)
        h = C.c_void_p()
        out = _u8p()
        n = C.c_size_t()
        err = self.L.orc_encode(body, X, Y, Nf, P, H, V, q, optimize, restart_interval,
                                C.byref(h), C.byref(out), C.byref(n))
        if err:
            return err, None, None
        img = self._image(h)
        jpg = C.string_at(out, n.value)
        self.L.orc_free(out)
        self.L.orc_image_free(h)
        return 0, jpg, img

    def encode_scans(self, samples, P=8, H=2, V=1, q=75, optimize=1, restart_interval=0, redefine=0, split=0):
        """EXTENSION: the image as one scan per component (decoder test input) -> (err, jpeg bytes)."""
        a = np.asarray(samples)
        Y, X = a.shape[:2]
        Nf = 1 if a.ndim == 2 else a.shape[2]
        out = _u8p()
        n = C.c_size_t()
        err = self.L.orc_encode_scans(pnm_body(a, P), X, Y, Nf, P, H, V, q, optimize, restart_interval, redefine, split,
                                      C.byref(out), C.byref(n))
        if err:
            return err, None
        jpg = C.string_at(out, n.value)
        self.L.orc_free(out)
        return 0, jpg

    # ---- primitives ---------------------------------------------------------------
    def lut(self):
        o = np.zeros(64, np.float32)
        self.L.orc_lut(o.ctypes.data_as(C.c_void_p))
        return o.reshape(8, 8)

    def _blk(self, fn, c):
        a = np.ascontiguousarray(c, np.float32).copy()
        fn(a.ctypes.data_as(C.c_void_p))
        return a

    def idct(self, c):
        return self._blk(self.L.orc_idct_block, c)

    def fdct(self, c):
        return self._blk(self.L.orc_fdct_block, c)

    def quantize(self, f, Q):
        f = np.ascontiguousarray(f, np.float32)
        Q = np.ascontiguousarray(Q, np.uint16)
        o = np.zeros(64, np.int32)
        self.L.orc_quantize_block(o.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p))
        return o

    def dequantize(self, c, Q):
        c = np.ascontiguousarray(c, np.int32)
        Q = np.ascontiguousarray(Q, np.uint16)
        o = np.zeros(64, np.float32)
        self.L.orc_dequantize_block(o.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p))
        return o

    def qtable(self, chroma, q):
        o = np.zeros(64, np.uint16)
        self.L.orc_set_qtable(o.ctypes.data_as(C.c_void_p), int(chroma), int(q))
        return o

    def default_table(self, Tc, Th):
        Lb = C.create_string_buffer(16)
        Vb = C.create_string_buffer(256)
        self.L.orc_default_table(Tc, Th, Lb, Vb)
        return np.frombuffer(Lb.raw, np.uint8).copy(), np.frombuffer(Vb.raw, np.uint8).copy()

    def make_codes(self, Lc, Vc):
        si = C.create_string_buffer(256)
        co = (C.c_uint16 * 256)()
        Lb = bytes(np.asarray(Lc, np.uint8))
        Vb = bytes(np.asarray(Vc, np.uint8)).ljust(256, b"\0")
        err = self.L.orc_make_codes(Lb, Vb, si, co)
        assert err == 0
        return np.frombuffer(si.raw, np.uint8).copy(), np.array(co[:], np.uint16)

    def adapt_table(self, freq):
        f = (C.c_size_t * 257)(*[int(v) for v in freq])
        Lb = C.create_string_buffer(16)
        Vb = C.create_string_buffer(256)
        self.L.orc_adapt_table(f, Lb, Vb)
        return np.frombuffer(Lb.raw, np.uint8).copy(), np.frombuffer(Vb.raw, np.uint8).copy()


class Reference:
    """The compiled reference (oracle/_ref).  Present in the build container and, as
    prebuilt files, on the GPU box; absent => tests that need it skip."""

    def __init__(self):
        if not have_ref():
            build()
        if not have_ref():
            raise RuntimeError("oracle/_ref is not built (no /root/reference here)")
        D = C.CDLL(os.path.join(HERE, "_ref", "libref_dec.so"))
        E = C.CDLL(os.path.join(HERE, "_ref", "libref_enc.so"))
        self.D, self.E = D, E
        D.ref_dec_run.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        E.ref_enc_run.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(_u8p), C.POINTER(C.c_size_t), C.c_int]
        E.ref_free.argtypes = [C.c_void_p]
        for lib in (D, E):
            lib.ref_tap_comp_info.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.POINTER(C.c_size_t)] * 2
            lib.ref_tap_coef.argtypes = [C.c_int]
            lib.ref_tap_coef.restype = C.POINTER(C.c_int32)
            lib.ref_tap_qtable.argtypes = [C.c_int]
            lib.ref_tap_qtable.restype = C.POINTER(C.c_uint16)
        D.ref_tap_frame.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        D.ref_tap_frame.restype = C.POINTER(C.c_float)
        D.ref_tap_frame_info.argtypes = [C.POINTER(C.c_int)] * 4

    def _taps(self, lib, img):
        for k in range(lib.ref_tap_ncomp()):
            ci = [C.c_int() for _ in range(4)]
            cs = [C.c_size_t() for _ in range(2)]
            lib.ref_tap_comp_info(k, *[C.byref(v) for v in ci], *[C.byref(v) for v in cs])
            bx, by = cs[0].value, cs[1].value
            coef = np.ctypeslib.as_array(lib.ref_tap_coef(k), shape=(bx * by, 64)).copy()
            img.comps.append(dict(id=ci[0].value, H=ci[1].value, V=ci[2].value, Tq=ci[3].value,
                                  bx=bx, by=by, coef=coef))
        img.qtables = np.stack([np.ctypeslib.as_array(lib.ref_tap_qtable(t), shape=(64,)).copy() for t in range(4)])
        if lib is self.E and hasattr(lib, "ref_tap_plane"):
            lib.ref_tap_plane.argtypes = [C.c_int]
            lib.ref_tap_plane.restype = C.POINTER(C.c_float)
            img.planes = [np.ctypeslib.as_array(lib.ref_tap_plane(k), shape=(c["by"] * 8, c["bx"] * 8)).copy()
                          for k, c in enumerate(img.comps)]

    def decode(self, jpg):
        """-> (err, samples array or None, image info).  Runs decoder.c:661 process_jpeg_stream."""
        with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None,
                                         suffix=".pnm", delete=False) as t:
            path = t.name
        try:
            err = self.D.ref_dec_run(jpg, len(jpg), path.encode(), 1)
            if err:
                return err, None, None
            with open(path, "rb") as f:
                arr, P = parse_pnm(f.read())
        finally:
            os.unlink(path)
        img = _Image()
        iv = [C.c_int() for _ in range(4)]
        self.D.ref_tap_frame_info(*[C.byref(v) for v in iv])
        img.P, img.X, img.Y, img.Nf = (v.value for v in iv)
        self._taps(self.D, img)
        sx, sy, nc = C.c_size_t(), C.c_size_t(), C.c_int()
        fp = self.D.ref_tap_frame(C.byref(sx), C.byref(sy), C.byref(nc))
        if fp:
            img.frame = np.ctypeslib.as_array(fp, shape=(sy.value, sx.value, nc.value)).copy()
        return 0, arr, img

    def encode(self, samples, P=8, H=2, V=1, q=75, optimize=1):
        """-> (err, jpeg bytes, image info).  Runs encoder.c:647 process_stream."""
        pnm = pnm_bytes(samples, P)
        out = _u8p()
        n = C.c_size_t()
        err = self.E.ref_enc_run(pnm, len(pnm), H, V, q, optimize, C.byref(out), C.byref(n), 1)
        jpg = C.string_at(out, n.value) if out else b""
        if out:
            self.E.ref_free(out)
        if err:
            return err, None, None
        img = _Image()
        self._taps(self.E, img)
        return 0, jpg, img

    def lut(self):
        o = np.zeros(64, np.float32)
        self.D.ref_dump_lut(o.ctypes.data_as(C.c_void_p))
        return o.reshape(8, 8)

    def _blk(self, fn, c):
        a = np.ascontiguousarray(c, np.float32).copy()
        fn(a.ctypes.data_as(C.c_void_p))
        return a

    def idct(self, c):
        return self._blk(self.D.ref_idct_block, c)

    def fdct(self, c):
        return self._blk(self.D.ref_fdct_block, c)

    def quantize(self, f, Q):
        f = np.ascontiguousarray(f, np.float32)
        Q = np.ascontiguousarray(Q, np.uint16)
        o = np.zeros(64, np.int32)
        self.D.ref_quantize_block(o.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p))
        return o
