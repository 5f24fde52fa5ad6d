"""Stage-level entry points (include/jpeg_amd.h: jpeg_hip_read_scan, jpeg_hip_decode_pixels,
jpeg_hip_encode_pixels, jpeg_hip_write_scan) against the oracle, through the C-ABI.

These are the four hot loops as a maintainer would call them from inside the reference's own
parse_format() / produce_codestream(): read_ecs (decoder.c:364-388), epilogue
(decoder.c:456-470), prologue (encoder.c:176-193), write_ecs_dry + write_ecs (encoder.c:525-587).  The
marker walk below plays the part of parse_format(); every stage result is compared bit for bit with what the
oracle holds at the same stage boundary."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from jpeg_amd.synth import synth_image

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def walk_scans(jpg):
    """A minimal parse_format(): -> list of (component ids, Td, Ta, ecs offset, offset of the ending marker)."""
    pos, scans = 2, []
    while pos + 4 <= len(jpg):
        assert jpg[pos] == 0xFF
        m = jpg[pos + 1]
        if m == 0xD9:
            break
        n = (jpg[pos + 2] << 8) | jpg[pos + 3]
        if m == 0xDA:
            ns = jpg[pos + 4]
            ids = [jpg[pos + 5 + 2 * j] for j in range(ns)]
            tt = [jpg[pos + 6 + 2 * j] for j in range(ns)]
            begin = pos + 2 + n
            end = begin
            while not (jpg[end] == 0xFF and jpg[end + 1] != 0 and not 0xD0 <= jpg[end + 1] <= 0xD7):
                end += 1
            scans.append((ids, [t >> 4 for t in tt], [t & 15 for t in tt], begin, end))
            pos = end
        else:
            pos += 2 + n
    return scans


def descs(hip, info, ids=None, Td=None, Ta=None, ri=0):
    fd = hip.frame_desc(info.X, info.Y, info.P, [c["H"] for c in info.comps], [c["V"] for c in info.comps],
                        [c["Tq"] for c in info.comps])
    if ids is None:
        return fd
    index = {c["id"]: k for k, c in enumerate(info.comps)}
    return fd, hip.scan_desc([index[i] for i in ids], Td, Ta, ri)


def tables(hip, info):
    return (hip.htables([info.htables[(0, th)] for th in range(4)]),
            hip.htables([info.htables[(1, th)] for th in range(4)]))


CASES = [
    # X, Y, Nf, P, H, V, q, optimize, restart interval
    (200, 136, 3, 8, 2, 2, 85, 1, 0),
    (333, 211, 3, 8, 2, 1, 70, 0, 7),
    (160, 120, 3, 8, 1, 1, 92, 1, 20),
    (257, 129, 1, 8, 1, 1, 80, 0, 0),
    (192, 160, 1, 12, 1, 1, 60, 1, 0),
    (1280, 720, 3, 8, 2, 2, 90, 1, 0),            # several workgroups per kernel
]


@pytest.mark.parametrize("case", CASES)
def test_every_stage_matches_the_oracle(hip, oracle, case):
    X, Y, Nf, P, H, V, q, opt, ri = case
    img = synth_image(X, Y, Nf, P, seed=31, range_safe=P > 8)
    err, jpg, einfo = oracle.encode(img, P=P, H=H, V=V, q=q, optimize=opt, restart_interval=ri)
    assert err == 0
    err, opx, dinfo = oracle.decode(jpg)
    assert err == 0
    (ids, Td, Ta, begin, end), = walk_scans(jpg)
    fd, sd = descs(hip, dinfo, ids, Td, Ta, ri)
    dc, ac = tables(hip, dinfo)

    # prologue: samples -> quantised coefficients
    err, coefs = hip.encode_pixels(fd, einfo.qtables, img)
    assert err == 0
    for k, c in enumerate(einfo.comps):
        assert np.array_equal(coefs[k], c["coef"]), "encode_pixels, component %d" % k

    # write_ecs[_dry]: coefficients -> entropy-coded bytes (and the adapted tables)
    wdc, wac = (hip.htables([None] * 4), hip.htables([None] * 4)) if opt else tables(hip, einfo)
    err, ecs = hip.write_scan(fd, sd, wdc, wac, opt, coefs)
    assert err == 0
    assert ecs == jpg[begin:end]
    if opt:
        for th in range(2 if Nf > 1 else 1):
            for tc, got in ((0, wdc), (1, wac)):
                Lw, Vw = einfo.htables[(tc, th)]
                nsym = int(np.sum(Lw))
                assert list(got[th].L) == list(Lw) and list(got[th].V)[:nsym] == list(Vw[:nsym])

    # read_ecs: entropy-coded bytes -> coefficients
    err, got, used = hip.read_scan(fd, sd, dc, ac, jpg[begin:])
    assert err == 0 and used == end - begin
    for k, c in enumerate(dinfo.comps):
        assert np.array_equal(got[k], c["coef"]), "read_scan, component %d" % k

    # epilogue: coefficients -> samples
    err, px = hip.decode_pixels(fd, dinfo.qtables, [got[k] for k in range(Nf)])
    assert err == 0
    assert np.array_equal(px, opx)


@pytest.mark.parametrize("split", [0, 1, 2])
def test_read_scan_on_frames_coded_in_several_scans(hip, oracle, split):
    """The call the reference makes once per SOS marker (decoder.c:593-601): each scan fills the buffers of
    its own components; together they are the frame the oracle decodes."""
    img = synth_image(264, 200, 3, 8, seed=32)
    for (H, V, ri) in [(2, 2, 0), (2, 1, 6), (1, 1, 9)]:
        err, jpg = oracle.encode_scans(img, P=8, H=H, V=V, q=85, optimize=1, restart_interval=ri, split=split)
        assert err == 0
        err, opx, info = oracle.decode(jpg)
        assert err == 0
        dc, ac = tables(hip, info)
        merged = {}
        for ids, Td, Ta, begin, end in walk_scans(jpg):
            fd, sd = descs(hip, info, ids, Td, Ta, ri)
            err, got, used = hip.read_scan(fd, sd, dc, ac, jpg[begin:])
            assert err == 0 and used == end - begin
            assert sorted(got) == sorted(sd.Cs[j] for j in range(sd.Ns))
            merged.update(got)
        for k, c in enumerate(info.comps):
            assert np.array_equal(merged[k], c["coef"])
        err, px = hip.decode_pixels(fd, info.qtables, [merged[k] for k in range(3)])
        assert err == 0 and np.array_equal(px, opx)


def test_decode_pixels_on_external_frames(hip, oracle):
    """epilogue() on frames of other encoders: four components (YCCK -> RGB, frame.c:196-217), 4:2:0 with
    libjpeg's tables, gray."""
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        man = json.load(f)["decode"]
    for name in ("d_pil_cmyk", "d_pil_420_rst_2mcu", "d_pil_gray", "d_pil_444_rst_row"):
        if name not in man:
            pytest.skip("fixture %s absent" % name)
        with open(os.path.join(HERE, "golden", man[name]["jpg_file"]), "rb") as f:
            jpg = f.read()
        err, opx, info = oracle.decode(jpg)
        assert err == 0
        fd = descs(hip, info)
        err, px = hip.decode_pixels(fd, info.qtables, [c["coef"] for c in info.comps])
        assert err == 0
        assert np.array_equal(px, opx), name


def test_stage_error_behaviour(hip, oracle):
    img = synth_image(96, 64, 3, 8, seed=33)
    err, jpg, einfo = oracle.encode(img, P=8, H=2, V=2, q=80, optimize=0)
    assert err == 0
    (ids, Td, Ta, begin, end), = walk_scans(jpg)
    fd, sd = descs(hip, einfo, ids, Td, Ta)
    dc, ac = tables(hip, einfo)
    coefs = [c["coef"] for c in einfo.comps]
    # data that ends without a marker: read_marker() meets the end of the file (io.c:197)
    err, got, used = hip.read_scan(fd, sd, dc, ac, jpg[begin:end])
    assert err == hip.RET_FAILURE_FILE_IO
    # ... but whatever was complete is in the buffers, as after a failed read_ecs()
    assert np.array_equal(got[1], coefs[1])
    # output buffer too small: the length needed comes back
    import ctypes as C
    n = C.c_size_t(0)
    small = np.zeros(16, np.uint8)
    ptrs = (C.c_void_p * 4)(*[c.ctypes.data for c in coefs], None)
    err = hip.lib().jpeg_hip_write_scan(0, C.byref(fd), C.byref(sd), dc, ac, 0, ptrs, small.ctypes.data, 16, C.byref(n))
    assert err == hip.RET_FAILURE_OVERFLOW_ERROR and n.value == end - begin
    # a coefficient the default tables have no code for (value_to_vlc() == -1, huffman.c:252)
    bad = [c.copy() for c in coefs]
    bad[0][5, 9] = 3000
    assert hip.write_scan(fd, sd, dc, ac, 0, bad)[0] == hip.JPEG_HIP_ERR_NO_CODE
    # |DC difference| > 2047: assert in the reference (encoder.c:447)
    bad = [c.copy() for c in coefs]
    bad[2][3, 0] = 2500
    assert hip.write_scan(fd, sd, dc, ac, 1, bad)[0] == hip.JPEG_HIP_ERR_DC_RANGE
    # frames the reference encoder never sets up
    odd = hip.frame_desc(96, 64, 8, [2, 2, 1], [1, 1, 1], [0, 1, 1])
    assert hip.encode_pixels(odd, einfo.qtables, img)[0] == hip.RET_FAILURE_FILE_UNSUPPORTED
    zero = einfo.qtables.copy()
    zero[1, 7] = 0
    assert hip.encode_pixels(fd, zero, img)[0] == hip.RET_FAILURE_FILE_UNSUPPORTED
    # a scan that names a component twice
    twice = hip.scan_desc([0, 0], [0, 0], [0, 0])
    assert hip.read_scan(fd, twice, dc, ac, jpg[begin:])[0] == hip.RET_FAILURE_FILE_UNSUPPORTED


def test_quantiser_division_against_the_compilers_own(hip):
    """quantize_block() is roundf(f / Q) in binary32 (imgproc.c:28).  The encode kernel hoists the divisor-only
    steps of hipcc's division expansion into the table and rounds by add-and-truncate (pixels.hip: quant_divide,
    round_half_away); jpeg_hip_selftest_divide runs both against `/` and roundf() on the device for every
    divisor 1 .. 65535, on and beside the rounding ties and on random numerators."""
    out = (C.c_ulonglong * 3)()
    for seed, per in ((1, 400), (2, 97)):
        assert hip.lib().jpeg_hip_selftest_divide(0, per, seed, out) == 0
        cases, badq, badr = (int(v) for v in out)
        assert cases > 50_000_000 and badq == 0 and badr == 0, (cases, badq, badr)


def test_colour_conversion_without_binary64_on_all_rgb_triples(hip):
    """frame_to_ycc (frame.c:154-186) evaluates in binary64; at precision 8 the encode kernel uses exact binary32
    integers and one correctly rounded division per value (jpeg_amd/csrc/ycc8.inc).  jpeg_hip_selftest_ycc8 runs
    both forms of it (one pixel, two pixels packed) against the binary64 expressions on the device for all 2^24
    (R, G, B); tests/test_ycc8_cpu.py runs the same text against the compiled reference on the host."""
    out = (C.c_ulonglong * 3)()
    assert hip.lib().jpeg_hip_selftest_ycc8(0, out) == 0
    assert [int(v) for v in out] == [1 << 24, 0, 0], [int(v) for v in out]


@pytest.mark.parametrize("X,Y,nc,P,H,V,q", [(83, 61, 3, 8, 2, 2, 75), (64, 48, 3, 8, 2, 1, 90), (40, 40, 3, 8, 1, 1, 50),
                                            (50, 37, 3, 8, 1, 2, 85), (72, 40, 1, 8, 1, 1, 80), (48, 32, 3, 12, 2, 2, 90),
                                            (33, 17, 1, 12, 1, 1, 90)])
def test_float_stage_boundaries_bit_for_bit(hip, oracle, X, Y, nc, P, H, V, q):
    """The two FLOAT stage boundaries of the reference that the fused kernels cross, tapped on the device and compared
    bit for bit with the oracle (which tests/test_oracle_vs_ref.py pins to the compiled reference at the same places):
    encode -- component[].frame_buffer behind frame_to_ycc + transform_frame_to_components (frame.c:84-186), padding
    included (quirk Q7); decode -- frame->data behind frame_to_rgb (frame.c:188-244), before roundf / clamp."""
    img = synth_image(X, Y, nc, P, seed=X + Y, range_safe=P > 8)
    err, jpg, oinfo = oracle.encode(img, P=P, H=H, V=V, q=q, optimize=1)
    assert err == 0
    e = hip.Encoder()
    e.configure(2, X, Y, nc, P, H, V, q, 1)
    planes = e.set_float_tap(True)
    e.run(e.upload([img, img]))
    assert e.sync()[0] == [0, 0]
    assert len(planes) == len(oinfo.planes)
    for k, want in enumerate(oinfo.planes):
        for f in range(2):
            got = planes[k][f].cpu().numpy()
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "plane %d of frame %d" % (k, f)
    e.set_float_tap(False)
    e.close()
    err, opx, dinfo = oracle.decode(jpg)
    assert err == 0
    d = hip.Decoder()
    d.prepare([jpg, jpg])
    tap = d.set_float_tap(True)
    out = d.run()
    assert d.sync() == [0, 0]
    oc = 1 if nc == 1 else 3
    want = dinfo.frame[:Y, :X, :oc]
    for f in range(2):
        got = tap[f].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), "frame %d" % f
    d.set_float_tap(False)
    d.close()


def test_float_frame_of_a_four_component_file(hip, oracle):
    """YCCK (frame.c:196-217): the first three channels of frame->data behind frame_to_rgb, Pillow-written CMYK golden."""
    from tests.conftest import golden_path
    jpg = open(golden_path("d_pil_cmyk.jpg"), "rb").read()
    err, opx, dinfo = oracle.decode(jpg)
    assert err == 0 and dinfo.Nf == 4
    d = hip.Decoder()
    d.prepare([jpg])
    tap = d.set_float_tap(True)
    d.run()
    assert d.sync() == [0]
    want = np.ascontiguousarray(dinfo.frame[:dinfo.Y, :dinfo.X, :3])
    assert np.array_equal(tap[0].cpu().numpy().view(np.uint32), want.view(np.uint32))
    d.close()


def test_single_stage_entry_points_against_the_oracles_primitives(hip, oracle):
    """jpeg_hip_dequantize_blocks / quantize_blocks / inverse_dct_blocks / forward_dct_blocks (imgproc.h:8-16 on the
    reference's flt_block / int_block arrays): every block bit for bit against the oracle's block primitives, which
    tests/test_oracle_vs_ref.py pins to the compiled reference's dequantize_block / quantize_block / idct / fdct."""
    L = hip.lib()
    rng = np.random.default_rng(5)
    n = 300
    Q = oracle.qtable(0, 37).astype(np.uint16)
    coef = rng.integers(-1200, 1200, (n, 64)).astype(np.int32)
    coef[rng.random((n, 64)) < 0.6] = 0
    f = np.zeros((n, 64), np.float32)
    assert L.jpeg_hip_dequantize_blocks(0, n, coef.ctypes.data, Q.ctypes.data, f.ctypes.data) == 0
    want = np.stack([oracle.dequantize(coef[b], Q) for b in range(n)])
    assert np.array_equal(f.view(np.uint32), want.view(np.uint32))
    for P in (8, 12):
        g = f.copy()
        assert L.jpeg_hip_inverse_dct_blocks(0, n, g.ctypes.data, P) == 0
        want = np.stack([oracle.idct(f[b]) + np.float32(1 << (P - 1)) for b in range(n)]).astype(np.float32)
        assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), P
        h = g.copy()
        assert L.jpeg_hip_forward_dct_blocks(0, n, h.ctypes.data, P) == 0
        want = np.stack([oracle.fdct((g[b] - np.float32(1 << (P - 1))).astype(np.float32)) for b in range(n)])
        assert np.array_equal(h.view(np.uint32), want.view(np.uint32)), P
        q = np.zeros((n, 64), np.int32)
        assert L.jpeg_hip_quantize_blocks(0, n, h.ctypes.data, Q.ctypes.data, q.ctypes.data) == 0
        want = np.stack([oracle.quantize(h[b], Q) for b in range(n)])
        assert np.array_equal(q, want), P
    zero = Q.copy()
    zero[9] = 0
    assert L.jpeg_hip_quantize_blocks(0, n, f.ctypes.data, zero.ctypes.data, coef.ctypes.data) == hip.RET_FAILURE_FILE_UNSUPPORTED
