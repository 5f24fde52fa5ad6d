"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the committed goldens.

Bar: bit-exact everywhere -- quantised coefficients, codestream bytes, decoded samples (the float
stages reproduce the reference's roundings exactly, so no tolerance is needed or allowed).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from jpeg_amd.synth import synth_image
from oracle.pyoracle import pnm_body
from tests.conftest import golden_path

pytestmark = pytest.mark.gpu


def md5(b):
    return hashlib.md5(b).hexdigest()


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _manifest():
    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")) as f:
        return json.load(f)


ENC = sorted(_manifest()["encode"].items())
DEC = sorted(_manifest()["decode"].items())


def _assert_coefs(hip_coefs, info):
    assert len(hip_coefs) == len(info.comps)
    for a, c in zip(hip_coefs, info.comps):
        assert a.shape == c["coef"].shape
        if not np.array_equal(a, c["coef"]):
            bad = np.argwhere(a != c["coef"])
            raise AssertionError("coefficients differ at %d places, first (block, idx) = %s: hip %d oracle %d" % (
                len(bad), bad[0], a[tuple(bad[0])], c["coef"][tuple(bad[0])]))


@pytest.mark.parametrize("name,g", ENC)
def test_encode_matches_golden_and_oracle(hip, oracle, name, g):
    img = synth_image(g["X"], g["Y"], g["ncomp"], g["P"], g["seed"], g["range_safe"])
    e = hip.Encoder()
    e.configure(1, g["X"], g["Y"], g["ncomp"], g["P"], g["H"], g["V"], g["q"], g["optimize"])
    e.run(e.upload([img]))
    st, ln = e.sync()
    assert st == [0]
    err, ojpg, oinfo = oracle.encode(img, P=g["P"], H=g["H"], V=g["V"], q=g["q"], optimize=g["optimize"])
    assert err == 0
    _assert_coefs(e.coefficients(0), oinfo)              # quantised coefficients, bit-exact
    jpg = e.jpeg(0, ln[0])
    assert len(jpg) == g["jpg_len"]
    assert jpg == ojpg                                   # Huffman bitstream + headers, bit-exact
    assert md5(jpg) == g["jpg_md5"]                      # = what the compiled reference produced
    e.close()


@pytest.mark.parametrize("name,g", ENC)
def test_decode_matches_golden_and_oracle(hip, oracle, name, g):
    if "jpg_file" in g:
        with open(golden_path(g["jpg_file"]), "rb") as f:
            jpg = f.read()
    else:
        img = synth_image(g["X"], g["Y"], g["ncomp"], g["P"], g["seed"], g["range_safe"])
        jpg = oracle.encode(img, P=g["P"], H=g["H"], V=g["V"], q=g["q"], optimize=g["optimize"])[1]
    assert md5(jpg) == g["jpg_md5"]
    d = hip.Decoder()
    d.prepare([jpg])
    out = d.run()
    assert d.sync() == [0]
    err, opx, oinfo = oracle.decode(jpg)
    assert err == 0
    _assert_coefs(d.coefficients(0), oinfo)              # entropy decode + DC un-prediction
    px = d.to_arrays(out)[0]
    assert px.shape == opx.shape
    assert np.array_equal(px, opx)                       # decoded samples, bit-exact
    assert md5(pnm_body(px, g["P"])) == g["dec_md5"]     # = what the compiled reference produced
    d.close()


@pytest.mark.parametrize("name,g", DEC)
def test_decode_external_streams(hip, oracle, name, g):
    with open(golden_path(g["jpg_file"]), "rb") as f:
        jpg = f.read()
    d = hip.Decoder()
    if g["err"] != 0:
        with pytest.raises(hip.JpegError) as ei:
            d.prepare([jpg])
        assert ei.value.code == g["err"]                 # e.g. progressive: RET_FAILURE_FILE_UNSUPPORTED
        return
    d.prepare([jpg])
    out = d.run()
    st = d.sync()
    assert st == [0]
    err, opx, oinfo = oracle.decode(jpg)
    _assert_coefs(d.coefficients(0), oinfo)
    px = d.to_arrays(out)[0]
    assert np.array_equal(px, opx)
    assert md5(pnm_body(px, g["P"])) == g["dec_md5"]
    d.close()


def test_batch_of_frames_with_different_tables(hip, oracle):
    """Motion-JPEG style batch: same geometry, per-frame optimised tables and quantisation."""
    imgs = [synth_image(208, 120, 3, 8, seed=50 + i, frame=i) for i in range(5)]
    jpgs = []
    for i, im in enumerate(imgs):
        err, j, _ = oracle.encode(im, P=8, H=2, V=1, q=60 + 8 * i, optimize=i % 2)
        assert err == 0
        jpgs.append(j)
    outs, st = hip.decode(jpgs)
    assert st == [0] * 5
    for j, o in zip(jpgs, outs):
        assert np.array_equal(o, oracle.decode(j)[1])
    # and the encoder on a batch
    out, st = hip.encode(imgs, P=8, H=2, V=1, q=85, optimize=1)
    assert st == [0] * 5
    for im, j in zip(imgs, out):
        assert j == oracle.encode(im, P=8, H=2, V=1, q=85, optimize=1)[1]


def test_restart_streams_from_the_oracle_extension(hip, oracle):
    """Streams WITH restart markers (oracle's restart extension, validated against the reference
    decoder in test_oracle_vs_ref) decode bit-exactly; intervals down to one MCU."""
    img = synth_image(176, 144, 3, 8, seed=77)
    for (H, V, ri) in [(2, 2, 1), (2, 2, 11), (1, 1, 22), (2, 1, 7)]:
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=88, optimize=1, restart_interval=ri)
        assert err == 0
        outs, st = hip.decode([jpg])
        assert st == [0]
        assert np.array_equal(outs[0], oracle.decode(jpg)[1])


def _dqt16(jpg, scale):
    """Rewrite every 8-bit DQT segment as a 16-bit one (Pq = 1) with the values scaled."""
    out, i = bytearray(jpg[:2]), 2
    while jpg[i + 1] != 0xDA:
        n = (jpg[i + 2] << 8) | jpg[i + 3]
        if jpg[i + 1] == 0xDB and n == 67 and (jpg[i + 4] >> 4) == 0:
            out += b"\xff\xdb\x00\x83" + bytes([0x10 | jpg[i + 4]])
            for v in jpg[i + 5:i + 69]:
                out += int(min(65535, v * scale)).to_bytes(2, "big")
        else:
            out += jpg[i:i + 2 + n]
        i += 2 + n
    return bytes(out + jpg[i:])


def test_quantisation_tables_of_16_bits_and_table_selectors(hip, oracle):
    """What the reference decoder takes beyond what its encoder writes (quirk Q4): DQT with Pq = 1 (entries
    far above 255: the product with the coefficient no longer fits 24 bits and the int -> float conversion of
    dequantize_block() rounds, imgproc.c:10-19), several tables in one DQT / DHT segment, SOF1."""
    img = synth_image(200, 136, 3, 8, seed=14)
    _, jpg, _ = oracle.encode(img, P=8, H=2, V=1, q=60, optimize=1)
    for scale in (1, 37, 700, 4000):
        wide = _dqt16(jpg, scale)
        err, opx, oinfo = oracle.decode(wide)
        assert err == 0 and int(oinfo.qtables.max()) == min(65535, int(oracle.decode(jpg)[2].qtables.max()) * scale)
        d = hip.Decoder()
        d.prepare([wide])
        out = d.run()
        assert d.sync() == [0]
        _assert_coefs(d.coefficients(0), oinfo)
        assert np.array_equal(d.to_arrays(out)[0], opx), scale
        d.close()
    # both DQT tables in ONE segment, all four DHT tables in ONE segment, SOF1 instead of SOF0
    segs, i = [], 2
    while jpg[i + 1] != 0xDA:
        n = (jpg[i + 2] << 8) | jpg[i + 3]
        segs.append((jpg[i + 1], jpg[i + 4:i + 2 + n]))
        i += 2 + n
    dqt = b"".join(b for m, b in segs if m == 0xDB)
    dht = b"".join(b for m, b in segs if m == 0xC4)
    sof = [b for m, b in segs if m == 0xC0][0]
    merged = (jpg[:2] + b"\xff\xdb" + (len(dqt) + 2).to_bytes(2, "big") + dqt + b"\xff\xc1" + (len(sof) + 2).to_bytes(2, "big") + sof +
              b"\xff\xc4" + (len(dht) + 2).to_bytes(2, "big") + dht + jpg[i:])
    err, opx, _ = oracle.decode(merged)
    assert err == 0 and np.array_equal(opx, oracle.decode(jpg)[1])
    outs, st = hip.decode([merged])
    assert st == [0] and np.array_equal(outs[0], opx)


def test_one_encoder_object_configured_again_and_again(hip, oracle):
    """configure() keeps the device buffers while their sizes stay the same (quality and table mode may change),
    sizes them anew when the batch or the geometry changes, and a refused configure() leaves the object
    unconfigured: every run must give the oracle's bytes."""
    e = hip.Encoder()
    imgs = [synth_image(200, 120, 3, 8, seed=80 + i) for i in range(3)]
    small = [synth_image(64, 40, 1, 8, seed=85)]
    for (batch, H, V, q, opt, ri) in [(imgs, 2, 2, 70, 1, 0), (imgs, 2, 2, 95, 0, 0), (imgs, 2, 2, 30, 1, 0),
                                       (imgs[:2], 2, 2, 30, 1, 0), (imgs, 2, 1, 80, 1, 0), (imgs, 2, 1, 80, 1, 4),
                                       (small, 1, 1, 85, 0, 0), (imgs, 2, 2, 70, 1, 0)]:
        a0 = batch[0]
        e.configure(len(batch), a0.shape[1], a0.shape[0], 1 if a0.ndim == 2 else 3, 8, H, V, q, opt, ri)
        e.run(e.upload(batch))
        st, ln = e.sync()
        assert st == [0] * len(batch)
        for f, im in enumerate(batch):
            assert e.jpeg(f, ln[f]) == oracle.encode(im, P=8, H=H, V=V, q=q, optimize=opt, restart_interval=ri)[1]
    with pytest.raises(hip.JpegError):
        e.configure(3, 200, 120, 2, 8, 1, 1, 70, 1, 0)       # two components: refused (encoder.c:147-148)
    with pytest.raises(hip.JpegError):
        e.run(e.upload(imgs))                                # ... and the object is not usable until configured
    e.close()


def test_frame_without_a_scan(hip, oracle):
    """EOI right behind the headers: the reference's epilogue() runs on the zero coefficients alloc_buffers()
    left (decoder.c:603-612, common.c:129) and writes a mid-grey image; so does this path."""
    img = synth_image(120, 72, 3, 8, seed=12)
    _, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=80, optimize=1)
    bare = jpg[:jpg.index(b"\xff\xda")] + b"\xff\xd9"
    err, opx, _ = oracle.decode(bare)
    assert err == 0
    outs, st = hip.decode([bare, bare])
    assert st == [0, 0]
    assert np.array_equal(outs[0], opx) and np.array_equal(outs[1], opx)
    # in one batch with frames that do have a scan: the odd one out is refused, the others decode
    outs, st = hip.decode([jpg, bare])
    assert st == [0, 0x1001] and np.array_equal(outs[0], oracle.decode(jpg)[1])


def test_restart_markers_the_header_does_not_announce(hip, oracle):
    """The reference never looks at Ri (quirk Q1, decoder.c:622-633): the markers decide.  A stream whose DRI
    is missing, or promises far fewer markers than come, decodes all the same (the segment tables are sized
    from DRI first, from the device's own count after that)."""
    imgs = [synth_image(320, 208, 3, 8, seed=70 + i) for i in range(3)]
    jpgs = []
    for im in imgs:
        err, jpg, _ = oracle.encode(im, P=8, H=2, V=1, q=85, optimize=1, restart_interval=1)
        assert err == 0
        jpgs.append(jpg)
    dri = jpgs[0].index(b"\xff\xdd")
    no_dri = jpgs[0][:dri] + jpgs[0][dri + 6:]
    wrong_dri = jpgs[1][:dri + 4] + b"\x7f\xff" + jpgs[1][dri + 6:]
    for batch in ([no_dri], [wrong_dri], [wrong_dri, jpgs[2][:dri + 4] + b"\x00\x63" + jpgs[2][dri + 6:]]):
        d = hip.Decoder()
        d.prepare(batch)
        out = d.run()
        assert d.sync() == [0] * len(batch)
        arrs = d.to_arrays(out)
        for i, j in enumerate(batch):
            err, opx, oinfo = oracle.decode(j)
            assert err == 0
            _assert_coefs(d.coefficients(i), oinfo)
            assert np.array_equal(arrs[i], opx)
        # the same prepared batch once more: the grown tables stay
        out = d.run()
        assert d.sync() == [0] * len(batch)
        assert np.array_equal(d.to_arrays(out)[0], oracle.decode(batch[0])[1])
        d.close()


def test_error_behaviour_matches_reference(hip, oracle):
    img = synth_image(64, 64, 3, 8, seed=3)
    _, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=80, optimize=1)
    # no EOI: the reference's read_marker() runs into end of file -> RET_FAILURE_FILE_IO
    outs, st = hip.decode([jpg[:-2]])
    assert st == [0x1000] and oracle.decode(jpg[:-2])[0] == 0x1000
    # truncated entropy-coded data but EOI present: decodes; the blocks of the unfinished MCU stay as far as they
    # were read (decoder.c:347-355, coeffs.c:222), the rest is mid grey.  The whole image is compared
    # (tests/test_gpu_damaged.py holds the many-cuts version).
    cut = jpg[:len(jpg) - 150] + b"\xff\xd9"
    outs, st = hip.decode([cut])
    err, opx, oinfo = oracle.decode(cut)
    assert st == [err] == [0]
    assert np.array_equal(outs[0], opx)
    # DC difference out of range: the reference assert()s (encoder.c:447); here a per-frame code
    bad = np.zeros((16, 64), np.uint16)
    bad[:, 32:] = 4095
    jp, st = hip.encode([bad], P=12, q=100, optimize=1)
    assert st == [hip.JPEG_HIP_ERR_DC_RANGE] and oracle.encode(bad, P=12, q=100, optimize=1)[0] == 0x5001
    # a category missing from the default tables: value_to_vlc() == -1
    noisy = ((np.arange(64 * 64).reshape(64, 64) * 2654435761 >> 7) % 4096).astype(np.uint16)
    oerr = oracle.encode(noisy, P=12, q=100, optimize=0)[0]
    jp, st = hip.encode([noisy], P=12, q=100, optimize=0)
    if oerr in (-1, 0x5001):
        assert st[0] in (hip.JPEG_HIP_ERR_NO_CODE, hip.JPEG_HIP_ERR_DC_RANGE)


def test_roundtrip_property_full_size_4k(hip, oracle):
    """BASELINE metric size (4K 4:2:0 q90): encode -> decode -> re-encode is a fixed point for the
    coefficients' entropy coding, and the codestream md5 equals the oracle's (one oracle run, ~1 s)."""
    img = synth_image(3840, 2160, 3, 8, seed=90)
    jpgs, st = hip.encode([img], P=8, H=2, V=2, q=90, optimize=0)
    assert st == [0]
    err, ojpg, oinfo = oracle.encode(img, P=8, H=2, V=2, q=90, optimize=0)
    assert err == 0 and md5(jpgs[0]) == md5(ojpg)
    d = hip.Decoder()
    d.prepare(jpgs)
    out = d.run()
    assert d.sync() == [0]
    _assert_coefs(d.coefficients(0), oinfo)              # decode(encode(coefs)) == coefs, 194 400 blocks
    px = d.to_arrays(out)[0]
    assert np.abs(px.astype(int) - img.astype(int)).max() < 64
    err, opx, _ = oracle.decode(jpgs[0])
    assert np.array_equal(px, opx)
    d.close()


def test_sync_paths_agree_on_a_multi_workgroup_stream(hip, oracle):
    """The three ways the speculative decoder can settle a workgroup's entry -- verified default
    seed, repair from the predecessor's carrier chain, serial walk -- give the same coefficients
    and samples as the oracle.  The stream spans dozens of chains workgroups."""
    img = synth_image(1920, 1088, 3, 8, seed=21)
    for (H, V, ri) in [(2, 2, 0), (2, 1, 0), (1, 1, 120)]:
        err, jpg, _ = oracle.encode(img, P=8, H=H, V=V, q=92, optimize=0, restart_interval=ri)
        assert err == 0
        err, opx, oinfo = oracle.decode(jpg)
        assert err == 0
        seen = []
        for dbg in (0, 1, 2):
            d = hip.Decoder()
            d.prepare([jpg, jpg])
            d.set_debug(dbg)
            out = d.run()
            assert d.sync() == [0, 0]
            seen.append(d.stats(1))
            _assert_coefs(d.coefficients(1), oinfo)
            for px in d.to_arrays(out):
                assert np.array_equal(px, opx)
            d.close()
        assert seen[0][0] == 0                           # fast path: no serial walk on a sane stream
        assert seen[1][1] > 0 or seen[1][0] != 0         # forced: workgroups really were repaired
        assert seen[2][0] != 0                           # forced: the serial walk really ran
        assert seen[0][2] == seen[1][2] == seen[2][2]    # same number of blocks decoded


def test_garbage_in_the_entropy_coded_data(hip, oracle):
    """Corrupt entropy-coded data (the true parse itself hits an undecodable code): whatever path the
    decoder takes, status and the whole image are the reference's."""
    img = synth_image(640, 480, 3, 8, seed=5)
    _, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=90, optimize=0)
    bad = bytearray(jpg)
    mid = len(bad) // 2
    bad[mid:mid + 64] = b"\xff\x00" * 32              # a long run of 1-bits: no code matches
    bad = bytes(bad)
    err, opx, oinfo = oracle.decode(bad)
    outs, st = hip.decode([bad])
    assert st == [err]
    if err == 0:
        assert np.array_equal(outs[0], opx)


def test_stray_markers_inside_the_entropy_coded_data(hip, oracle):
    """A marker other than RSTn ends read_ecs() (io.c:257-268); parse_format() then goes on from it
    (decoder.c:487-657): TEM and segments with a length pass, the bytes up to the next marker are skipped
    (io.c:196-210), EOI ends the image with what was decoded.  Same status and same image here."""
    img = synth_image(320, 240, 3, 8, seed=8)
    _, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=85, optimize=1)
    sos = jpg.index(b"\xff\xda")
    start = sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])
    mid = start + (len(jpg) - start) // 2
    for stray in (b"\xff\x01", b"\xff\xfe\x00\x04ab", b"\xff\xe1\x00\x02"):
        bad = jpg[:mid] + stray + jpg[mid:]
        err, opx, oinfo = oracle.decode(bad)
        outs, st = hip.decode([bad])
        assert err == 0 and st == [0], stray
        assert np.array_equal(outs[0], opx), stray
    # new quantisation tables behind the scan would change the image (the reference dequantises at EOI,
    # decoder.c:456-470), markers the reference does not know end in its "unhandled marker": both refused
    dqt = jpg[jpg.index(b"\xff\xdb"):][:69]
    changed = dqt[:5] + bytes([dqt[5] ^ 3]) + dqt[6:]
    assert hip.decode([jpg[:mid] + changed + jpg[mid:]])[1] == [0x1001]
    bad = jpg[:mid] + b"\xff\xe9\x00\x02" + jpg[mid:]            # APP9
    assert hip.decode([bad])[1] == [0x1001] and oracle.decode(bad)[0] == 0x1001
    # nothing but data behind the stray marker: read_marker() meets the end of the file
    bad = jpg[:mid] + b"\xff\x01" + jpg[mid:-2]
    assert hip.decode([bad])[1] == [0x1000] and oracle.decode(bad)[0] == 0x1000


def _cut_intervals(jpg, which, keep=0.55):
    """Drop the tail of the restart intervals listed in `which` (their data then ends inside an MCU)."""
    sos = jpg.index(b"\xff\xda")
    start = sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])
    marks = [i for i in range(start, len(jpg) - 1) if jpg[i] == 0xFF and 0xD0 <= jpg[i + 1] <= 0xD7]
    bounds = [start] + [m + 2 for m in marks]          # first data byte of every interval
    ends = marks + [len(jpg) - 2]                      # the marker (or EOI) behind it
    out = bytearray(jpg[:start])
    for k, (b, e) in enumerate(zip(bounds, ends)):
        n = e - b
        if k in which:
            n = max(2, int(n * keep))
            while jpg[b + n - 1] == 0xFF:              # never end on the first half of FF 00
                n -= 1
        out += jpg[b:b + n] + jpg[e:e + 2]
    return bytes(out)


@pytest.mark.parametrize("sync_debug", [0, 2])
def test_intervals_that_end_inside_an_mcu(hip, oracle, sync_debug):
    """Truncated restart intervals: the reference does not count the unfinished MCU (decoder.c:376-381),
    so the next interval's blocks go where its blocks were stored.  On the GPU both are stored by
    unrelated threads; k_huff_fixup settles the order.  Cut intervals whose successor is complete give
    exactly the reference's coefficients and samples."""
    img = synth_image(256, 192, 3, 8, seed=77)
    err, jpg, _ = oracle.encode(img, P=8, H=2, V=2, q=85, optimize=1, restart_interval=3)
    assert err == 0
    bad = _cut_intervals(jpg, {1, 4, 5 + 2, 20, 41, 42 + 9})
    err, opx, oinfo = oracle.decode(bad)
    assert err == 0
    d = hip.Decoder()
    d.prepare([bad])
    d.set_debug(sync_debug)
    out = d.run()
    assert d.sync() == [0]
    _assert_coefs(d.coefficients(0), oinfo)
    assert np.array_equal(d.to_arrays(out)[0], opx)
    d.close()


@pytest.mark.parametrize("split", [0, 1, 2])
def test_frames_coded_in_several_scans(hip, oracle, split):
    """Scans of one component (decoder.c:274-302) and of some of the components (decoder.c:303-358 with
    Ns < Nf), written by the oracle's extension writer (judged by the reference decoder in
    tests/golden/make_golden.py and test_oracle_vs_ref): a batch large enough for several workgroups per
    scan, with and without restart intervals, on every synchronisation path."""
    imgs = [synth_image(416, 272, 3, 8, seed=90 + i, frame=i) for i in range(3)]
    for (H, V, ri, opt) in [(2, 2, 0, 1), (2, 1, 5, 0), (1, 1, 8, 1), (1, 2, 3, 1)]:
        jpgs = []
        for im in imgs:
            err, j = oracle.encode_scans(im, P=8, H=H, V=V, q=88, optimize=opt, restart_interval=ri, split=split)
            assert err == 0
            jpgs.append(j)
        for dbg in (0, 1, 2):
            d = hip.Decoder()
            d.set_debug(dbg)
            d.prepare(jpgs)
            out = d.run()
            assert d.sync() == [0] * len(jpgs)
            arrs = d.to_arrays(out)
            for i, j in enumerate(jpgs):
                err, opx, oinfo = oracle.decode(j)
                assert err == 0
                _assert_coefs(d.coefficients(i), oinfo)
                assert np.array_equal(arrs[i], opx)
            d.close()
    # frames of one batch must share the scan structure
    a = oracle.encode_scans(imgs[0], P=8, H=2, V=2, q=80, split=split)[1]
    b = oracle.encode_scans(imgs[1], P=8, H=2, V=2, q=80, split=(split + 1) % 3)[1]
    outs, st = hip.decode([a, b])
    assert st == [0, 0x1001]                              # RET_FAILURE_FILE_UNSUPPORTED for the odd one out
    assert np.array_equal(outs[0], oracle.decode(a)[1])


def test_motion_jpeg_stream_in_batches(hip, oracle):
    """A concatenated stream (incl. DHT-less frames that rely on the implicit tables, common.c:90-99)
    is split by the C++ splitter and decoded in batches; every frame equals the oracle's decode."""
    from jpeg_amd import mjpeg
    with open(golden_path("d_mjpeg_nodht_422.jpg"), "rb") as f:
        nodht = f.read()
    err, ref_nodht, info = oracle.decode(nodht)
    assert err == 0
    X, Y = ref_nodht.shape[1], ref_nodht.shape[0]
    frames = [nodht]
    for i in range(6):
        img = synth_image(X, Y, 3, 8, seed=900 + i, frame=i)
        err, j, _ = oracle.encode(img, P=8, H=2, V=1, q=85, optimize=i % 2)
        assert err == 0
        frames.append(j)
    got = mjpeg.split(b"junk" + b"".join(frames))
    assert got == frames
    lo, hi = mjpeg.shard(len(got), 0, 1)
    k = 0
    sums = []
    for out, st in mjpeg.decode_batches(got[lo:hi], batch=3):
        assert all(s == 0 for s in st)
        arr = out.cpu().numpy()
        for row in arr:
            want = oracle.decode(frames[k])[1]
            assert np.array_equal(row.reshape(want.shape), want)
            sums.append(mjpeg.frame_checksum(want.tobytes()))
            k += 1
    assert k == len(frames)
    tot = mjpeg.gather_stats({"frames": k, "bytes": 0, "checksum": mjpeg.combine_checksums(sums)})
    assert tot["frames"] == len(frames)


def test_restart_marker_encode_extension(hip, oracle, reference):
    """DRI / RSTn on the encoder side are an extension (the reference writes none).
    The GPU encoder's stream equals the oracle's byte for byte, and the REFERENCE decoder (the judge of
    this extension) decodes it to the same samples as the stream without restart markers."""
    img = synth_image(200, 136, 3, 8, seed=11)
    gray = synth_image(120, 72, 1, 8, seed=12)
    cases = [(img, 2, 2, 1, 0), (img, 2, 2, 5, 1), (img, 2, 1, 13, 0), (img, 1, 1, 25, 1), (img, 2, 2, 1000, 0),
             (gray, 1, 1, 4, 1)]
    for im, H, V, ri, opt in cases:
        jpgs, st = hip.encode([im, im], P=8, H=H, V=V, q=80, optimize=opt, restart_interval=ri)
        assert st == [0, 0]
        err, oj, _ = oracle.encode(im, P=8, H=H, V=V, q=80, optimize=opt, restart_interval=ri)
        assert err == 0
        assert jpgs[0] == oj and jpgs[1] == oj
        err, plain, _ = oracle.encode(im, P=8, H=H, V=V, q=80, optimize=opt)
        rerr, rpx, _ = reference.decode(jpgs[0])
        assert rerr == 0 and np.array_equal(rpx, reference.decode(plain)[1])
        outs, st = hip.decode(jpgs)
        assert st == [0, 0] and np.array_equal(outs[0], rpx)


def test_double_buffered_stream_equals_batches(hip, oracle):
    """mjpeg.decode_stream (two decoders taking turns, uploads from a helper thread, decode on a stream of
    its own) yields what decode_batches yields, batch after batch -- also when a decoder is prepared again
    with a smaller and then a larger batch (its buffers only grow)."""
    from jpeg_amd import mjpeg
    frames = []
    for i in range(11):
        img = synth_image(160, 96, 3, 8, seed=300 + i, frame=i)
        err, j, _ = oracle.encode(img, P=8, H=2, V=1, q=60 + 3 * i, optimize=i % 2)
        assert err == 0
        frames.append(j)
    a = [(o.cpu().numpy().copy(), st) for o, st in mjpeg.decode_batches(frames, batch=4)]
    b = [(o.cpu().numpy().copy(), st) for o, st in mjpeg.decode_stream(frames, batch=4)]
    assert len(a) == len(b) == 3
    for (oa, sa), (ob, sb) in zip(a, b):
        assert sa == sb and all(s == 0 for s in sa)
        assert np.array_equal(oa, ob)
    d = hip.Decoder()
    for part in (frames[:6], frames[6:8], frames):
        d.prepare(part)
        out = d.run()
        assert d.sync() == [0] * len(part)
        for k, px in enumerate(d.to_arrays(out)):
            assert np.array_equal(px, oracle.decode(part[k])[1])
    d.close()


def test_frames_in_page_locked_memory(hip, oracle):
    """jpeg_hip_decoder_prepare_pinned: the frames go to the device in one DMA from the caller's buffer, at whatever
    offsets they have there.  Same samples and statuses as the staged path -- whole stream and batches of odd
    sizes, a frame with a stray marker that sync() has to cut (the caller's memory stays as it was), a broken
    frame in the middle, a single frame, and one batch of 1, 2, 5 frames through mjpeg.decode_stream."""
    from jpeg_amd import mjpeg
    frames = []
    for i in range(9):
        img = synth_image(200, 120, 3, 8, seed=500 + i, frame=i)
        err, j, _ = oracle.encode(img, P=8, H=2, V=2, q=55 + 4 * i, optimize=i % 2, restart_interval=(0, 5)[i % 2])
        assert err == 0
        frames.append(j)
    j = frames[3]
    marks = [k for k in range(len(j) - 1) if j[k] == 0xFF and 0xD0 <= j[k + 1] <= 0xD7]
    at = marks[4] - 3
    at += j[at - 1] == 0xFF
    frames[3] = j[:at] + b"\xff\xe1\x00\x05abc" + j[at:]        # stray APP1, restart intervals behind it
    frames[6] = frames[6][:40]                                # no frame header left: parse error
    want = [oracle.decode(f) for f in frames]
    pf = hip.PinnedFrames.from_frames(frames)
    before = bytes(pf.buf.view())
    for lo, hi in ((0, 9), (2, 5), (3, 4), (5, 9), (0, 1)):
        d = hip.Decoder()
        d.prepare(pf[lo:hi])
        out = d.run()
        st = d.sync()
        px = d.to_arrays(out)
        for k in range(lo, hi):
            assert st[k - lo] == want[k][0], (lo, hi, k)
            if want[k][0] == 0:
                assert np.array_equal(px[k - lo], want[k][1]), (lo, hi, k)
        # a second run of the same preparation (the cut frame is already cut)
        out = d.run()
        assert d.sync() == st
        d.close()
    assert bytes(pf.buf.view()) == before
    good = hip.PinnedFrames.from_frames([f for k, f in enumerate(frames) if k != 6])
    a = [(o.cpu().numpy().copy(), st) for o, st in mjpeg.decode_batches([f for k, f in enumerate(frames) if k != 6], batch=3)]
    for batch in (3, 1, 8):
        b = [(o.cpu().numpy().copy(), st) for o, st in mjpeg.decode_stream(good, batch=batch)]
        assert np.array_equal(np.concatenate([o for o, _ in a]), np.concatenate([o for o, _ in b]))
        assert sum((st for _, st in b), []) == [0] * 8
    pf.buf.close()
    good.buf.close()


def test_pinned_frames_scattered_over_memory(hip, oracle):
    """jpeg_hip_decoder_prepare_pinned with frames in different page-locked buffers, far apart or in descending
    order: a span that is mostly holes falls back to the staged path; either way the samples are the oracle's."""
    import ctypes as C
    frames = []
    for i in range(3):
        img = synth_image(120, 88, 3, 8, seed=700 + i)
        err, j, _ = oracle.encode(img, P=8, H=2, V=1, q=70 + i, optimize=1)
        assert err == 0
        frames.append(j)
    bufs = [hip.PinnedBuffer(64 << 20), hip.PinnedBuffer(len(frames[1]) + 100), hip.PinnedBuffer(1 << 20)]
    offs = [(48 << 20) + 3, 7, 4096]
    for b, o, j in zip(bufs, offs, frames):
        C.memmove(b.addr + o, j, len(j))
    for order in ((0, 1, 2), (2, 1, 0), (1, 0), (0,)):
        n = len(order)
        ptrs = (C.c_void_p * n)(*[bufs[k].addr + offs[k] for k in order])
        lens = (C.c_size_t * n)(*[len(frames[k]) for k in order])
        d = hip.Decoder()
        assert d.L.jpeg_hip_decoder_prepare_pinned(d.h, n, ptrs, lens) == 0
        d.n = n
        d.info = hip.FrameInfo()
        d.L.jpeg_hip_decoder_info(d.h, C.byref(d.info))
        d.frame_bytes = d.L.jpeg_hip_decoder_frame_bytes(d.h)
        out = d.run()
        assert d.sync() == [0] * n
        for px, k in zip(d.to_arrays(out), order):
            assert np.array_equal(px, oracle.decode(frames[k])[1]), order
        d.close()
    for b in bufs:
        b.close()


@pytest.mark.parametrize("P", [13, 14, 15, 16])
def test_sample_precisions_beyond_12_bits(hip, oracle, P):
    """frame.c:270-275: any maxval up to 65535, so P up to 16.  The reference's own assert on |DC| <= 2047
    (encoder.c:447) leaves it low-contrast images around mid grey at such precisions; their coefficients fit the
    16 bits this path stores (a frame whose do not is refused, not stored wrapped).  Codestream bytes, coefficients
    and decoded samples against the oracle, both table modes, gray and 4:2:0, restart intervals."""
    from tests.damaged import deep_image
    for nc, H, V in ((1, 1, 1), (3, 2, 2), (3, 1, 1)):
        img = deep_image(P, nc, seed=P + nc)
        for q, opt, ri in ((30, 1, 0), (50, 0, 0), (50, 1, 7)):
            err, ojpg, oinfo = oracle.encode(img, P=P, H=H, V=V, q=q, optimize=opt, restart_interval=ri)
            assert err == 0, (P, nc, q, opt)
            e = hip.Encoder()
            e.configure(1, img.shape[1], img.shape[0], nc, P, H, V, q, opt, ri)
            e.run(e.upload([img]))
            st, ln = e.sync()
            assert st == [0], (P, nc, q, opt)
            _assert_coefs(e.coefficients(0), oinfo)
            assert md5(e.jpeg(0, ln[0])) == md5(ojpg), (P, nc, q, opt, ri)
            e.close()
            oerr, opx, odinfo = oracle.decode(ojpg)
            assert oerr == 0
            d = hip.Decoder()
            d.prepare([ojpg, ojpg])
            out = d.run()
            assert d.sync() == [0, 0]
            _assert_coefs(d.coefficients(1), odinfo)
            for px in d.to_arrays(out):
                assert np.array_equal(px, opx), (P, nc, q, opt, ri)
            d.close()
    # an image the reference refuses (DC out of range): the same error here
    bright = np.full((32, 64), (1 << P) - 1, np.uint16)
    assert oracle.encode(bright, P=P, H=1, V=1, q=90, optimize=1)[0] == 0x5001
    jp, st = hip.encode([bright], P=P, q=90, optimize=1)
    assert st == [hip.JPEG_HIP_ERR_DC_RANGE]


def test_markers_the_reference_rejects(hip, oracle):
    """APP9 / APP10 / APP15 / DNL end the reference's parse_format() with "unhandled marker"
    (decoder.c:654-656): the default is the same status; set_tolerant() skips such segments."""
    with open(golden_path("e_420_q75_o1.jpg"), "rb") as f:
        jpg = f.read()
    err, opx, _ = oracle.decode(jpg)
    assert err == 0
    for marker in (0xE9, 0xEA, 0xEF):
        odd = jpg[:2] + bytes([0xFF, marker, 0x00, 0x06]) + b"abcd" + jpg[2:]
        rerr = oracle.decode(odd)[0]
        assert rerr == hip.RET_FAILURE_FILE_UNSUPPORTED
        d = hip.Decoder()
        with pytest.raises(hip.JpegError) as e:
            d.prepare([odd])
        assert e.value.code == rerr
        d.set_tolerant(True)
        d.prepare([odd])
        out = d.run()
        assert d.sync() == [0]
        assert np.array_equal(d.to_arrays(out)[0], opx)
        d.close()


def test_randomised_cases(hip, oracle):
    """A bounded run of tools/fuzz.py: random geometry, sampling, precision, quality, table mode, restart
    interval and batch size; codestream bytes, coefficients and samples identical to the oracle's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz.py"), "60", "7"], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_side_image_generator_matches_the_host_one(hip):
    """bench.py makes its batches of distinct frames on the GPU (jpeg_amd.synth.synth_body_torch); the recipe is
    integer-only, so the bodies must be the bytes of synth_image() -- the generator every parity test uses."""
    import torch
    from jpeg_amd.synth import synth_body_torch
    from oracle.pyoracle import pnm_body
    for X, Y, nc, P, seed, safe, fr in [(200, 120, 3, 8, 90, False, 7), (96, 64, 1, 12, 3, True, 2), (64, 40, 3, 12, 9, False, 1)]:
        want = pnm_body(synth_image(X, Y, nc, P, seed, safe, frame=fr), P)
        got = synth_body_torch(X, Y, nc, P, seed, safe, frame=fr, device="cuda:0").cpu().numpy().tobytes()
        assert got == want, (X, Y, nc, P)


@pytest.mark.parametrize("P,opt", [(8, 0), (8, 1), (12, 1)])
def test_noise_at_top_quality_outgrows_the_stream_buffers(hip, oracle, P, opt):
    """Uniform noise at q = 100 codes to MORE bytes than the samples take: the encoder's unstuffed stream and
    output slots are sized for ordinary images, sync() grows them and codes the batch again.  The reference
    writes to a FILE* without limit, so the bytes must simply be the oracle's.  Second frame of the batch: an
    ordinary image (its stream must survive the second pass unchanged)."""
    rng = np.random.default_rng(1000 + P)
    X, Y = 1280, 1024
    if P == 8:
        noise = rng.integers(0, 256, (Y, X)).astype(np.uint8)
    else:
        # full-contrast noise with every block's mean at mid-grey: |DC difference| stays below 2047 (quirk Q3)
        n = rng.integers(-1800, 1800, (Y // 8, 8, X // 8, 8))
        n = n - n.mean(axis=(1, 3), keepdims=True).astype(np.int64)
        noise = (2048 + n).reshape(Y, X).clip(0, 4095).astype(np.uint16)
    calm = synth_image(X, Y, 1, 8, seed=3)
    if P == 12:
        calm = (2048 + calm.astype(np.int32) - 128).astype(np.uint16)    # DC within +-2047 at Q = 1 as well
    nblk = (X // 8) * (Y // 8)
    e = hip.Encoder(0)
    try:
        for rounds in range(2):                            # the second configure() keeps the grown buffers
            e.configure(2, X, Y, 1, P, 1, 1, 100, opt)
            e.run(e.upload([noise, calm]))
            st, ln = e.sync()
            assert st == [0, 0]
            for f, im in enumerate((noise, calm)):
                err, want, _ = oracle.encode(im, P=P, H=1, V=1, q=100, optimize=opt)
                assert err == 0
                assert e.jpeg(f, ln[f]) == want
            if P == 8 and opt == 0:
                # more than configure() reserves (api.hip: raw + 4 bytes per block + 64 KiB): the second pass ran
                assert ln[0] > X * Y + nblk * 4 + 65536 + 4096
    finally:
        e.close()
    # the stage-level call goes through the same growth
    if P == 8 and opt == 0:
        err, want, info = oracle.encode(noise, P=P, H=1, V=1, q=100, optimize=opt)
        fd = hip.frame_desc(X, Y, P, [1], [1], [0])
        sd = hip.scan_desc([0], [0], [0])
        dc = hip.htables([oracle.default_table(0, 0)] + [None] * 3)
        ac = hip.htables([oracle.default_table(1, 0)] + [None] * 3)
        err, ecs = hip.write_scan(fd, sd, dc, ac, 0, [info.comps[0]["coef"]])
        assert err == 0 and ecs in want


def _scan_segments(jpg):
    """(start of SOS marker, end of its entropy-coded data) of every scan of a file"""
    out = []
    i = 2
    while i + 3 < len(jpg):
        assert jpg[i] == 0xFF, i
        m = jpg[i + 1]
        if m == 0xD9:
            break
        ln = (jpg[i + 2] << 8) | jpg[i + 3]
        if m != 0xDA:
            i += 2 + ln
            continue
        j = i + 2 + ln
        while not (jpg[j] == 0xFF and jpg[j + 1] != 0 and not 0xD0 <= jpg[j + 1] <= 0xD7):
            j += 1
        out.append((i, j))
        i = j
    return out


def test_a_dozen_scans(hip, oracle):
    """The reference reads as many scans as the file holds (decoder.c:593-601); the decoder keeps a pass per scan,
    up to 32.  Luma coded ten times over (scans of a second and a third image taking turns) behind the three
    per-component scans of the first: 13 scans; 33 scans are refused as unsupported."""
    imgs = [synth_image(176, 112, 3, 8, seed=61 + i) for i in range(3)]
    files = []
    for im in imgs:
        err, j = oracle.encode_scans(im, P=8, H=2, V=2, q=80, optimize=0, restart_interval=0, redefine=0, split=0)
        assert err == 0
        files.append(j)
    segs = [_scan_segments(j) for j in files]
    ja, sa = files[0], segs[0]
    head, tail = ja[:sa[-1][1]], ja[sa[-1][1]:]
    luma = [files[k][segs[k][0][0]:segs[k][0][1]] for k in (1, 2)]
    jpg = head + b"".join(luma[i % 2] for i in range(10)) + tail
    assert len(_scan_segments(jpg)) == 13
    oerr, opx, oinfo = oracle.decode(jpg)
    assert oerr == 0
    d = hip.Decoder()
    try:
        d.prepare([jpg, jpg])                            # (the frames of a batch have their scans in common)
        out = d.run()
        assert d.sync() == [0, 0]
        _assert_coefs(d.coefficients(1), oinfo)
        px = d.to_arrays(out)
        assert np.array_equal(px[0], opx) and np.array_equal(px[1], opx)
        assert not np.array_equal(opx, oracle.decode(ja)[1])
        # the last luma scan is image 2's (i = 9 is odd: luma[1]): luma of that image, chroma of the first
        many = head + b"".join(luma[i % 2] for i in range(30)) + tail
        assert len(_scan_segments(many)) == 33
        with pytest.raises(hip.JpegError):
            d.prepare([many])
    finally:
        d.close()


def test_eight_huffman_tables_in_one_scan(hip, oracle):
    """Four components with a DC and an AC table each (tests/damaged.py: table_per_component on the CMYK golden):
    the most look-up tables the Huffman kernels hold in LDS (k_huff_write: 68 KB with its block buffers).  Alone,
    and in a batch beside the file it was made from (two tables)."""
    from tests.conftest import golden_path
    from tests.damaged import table_per_component
    with open(golden_path("d_pil_cmyk.jpg"), "rb") as fh:
        jpg = fh.read()
    eight = table_per_component(jpg)
    oerr, opx, oinfo = oracle.decode(eight)
    assert oerr == 0 and oinfo.Nf == 4
    d = hip.Decoder()
    try:
        for batch in ([eight], [jpg, eight, eight, jpg]):
            d.prepare(batch)
            out = d.run()
            assert d.sync() == [0] * len(batch)
            for k in range(len(batch)):
                _assert_coefs(d.coefficients(k), oinfo)
            for px in d.to_arrays(out):
                assert np.array_equal(px, opx)
    finally:
        d.close()


def test_components_coded_more_than_once(hip, oracle):
    """decoder.c:593-601 reads whatever scans come; read_block() stores every coefficient of a block it reads
    (coeffs.c:222-229), so a component that is coded again is simply written again.  Files made by splicing scans
    of a second image (same size, same tables) into the first: luma twice, all components again behind the
    per-component scans, a group of two behind them, restart intervals on and off."""
    a = synth_image(176, 112, 3, 8, seed=41)
    b = synth_image(176, 112, 3, 8, seed=42)
    for ri in (0, 5):
        err, ja = oracle.encode_scans(a, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri, redefine=0, split=0)
        err2, jb = oracle.encode_scans(b, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri, redefine=0, split=0)
        err3, jb1 = oracle.encode_scans(b, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri, redefine=0, split=1)
        err4, jfull, _ = oracle.encode(b, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri)
        assert err == err2 == err3 == err4 == 0
        sa, sb, sb1, sf = _scan_segments(ja), _scan_segments(jb), _scan_segments(jb1), _scan_segments(jfull)
        assert len(sa) == 3 and len(sb) == 3 and len(sf) == 1
        head, tail = ja[:sa[-1][1]], ja[sa[-1][1]:]
        files = {
            "luma twice": head + jb[sb[0][0]:sb[0][1]] + tail,
            "all again": head + jfull[sf[0][0]:sf[0][1]] + tail,
            "a group again": head + jb1[sb1[-1][0]:sb1[-1][1]] + tail,
            "chroma first, luma twice": ja[:sa[0][0]] + ja[sa[1][0]:sa[2][1]] + ja[sa[0][0]:sa[0][1]] + jb[sb[0][0]:sb[0][1]] + tail,
        }
        for name, jpg in files.items():
            oerr, opx, oinfo = oracle.decode(jpg)
            assert oerr == 0, name
            d = hip.Decoder()
            d.prepare([jpg, jpg])
            out = d.run()
            assert d.sync() == [0, 0], (name, ri)
            _assert_coefs(d.coefficients(1), oinfo)
            for px in d.to_arrays(out):
                assert np.array_equal(px, opx), (name, ri)
            d.close()
        assert not np.array_equal(oracle.decode(files["luma twice"])[1], oracle.decode(ja)[1])
        # a first scan with every component and more scans behind it: prepare() leaves the end of such a scan to
        # the device, sync() finds the SOS behind it and decodes the batch again scan by scan
        err, jfa, _ = oracle.encode(a, P=8, H=2, V=2, q=80, optimize=0, restart_interval=ri)
        sfa = _scan_segments(jfa)
        again = {
            "all, then luma": jfa[:sfa[0][1]] + jb[sb[0][0]:sb[0][1]] + jfa[sfa[0][1]:],
            "all twice": jfa[:sfa[0][1]] + jfull[sf[0][0]:sf[0][1]] + jfa[sfa[0][1]:],
        }
        for name, jpg in again.items():
            oerr, opx, oinfo = oracle.decode(jpg)
            assert oerr == 0, name
            for pinned in (False, True):
                d = hip.Decoder()
                src = hip.PinnedFrames.from_frames([jpg, jpg, jpg]) if pinned else [jpg, jpg, jpg]
                d.prepare(src)
                out = d.run()
                assert d.sync() == [0, 0, 0], (name, ri, pinned)
                _assert_coefs(d.coefficients(2), oinfo)
                for px in d.to_arrays(out):
                    assert np.array_equal(px, opx), (name, ri)
                # the same decoder on ordinary frames afterwards
                d.prepare([jfa])
                out = d.run()
                assert d.sync() == [0]
                assert np.array_equal(d.to_arrays(out)[0], oracle.decode(jfa)[1])
                d.close()
        # gray, coded twice; and a batch in which only one frame has the second scan (that frame is refused)
        ga, gb = synth_image(96, 64, 1, 8, seed=5), synth_image(96, 64, 1, 8, seed=6)
        _, jga, _ = oracle.encode(ga, P=8, H=1, V=1, q=75, optimize=0, restart_interval=ri)
        _, jgb, _ = oracle.encode(gb, P=8, H=1, V=1, q=75, optimize=0, restart_interval=ri)
        sga, sgb = _scan_segments(jga), _scan_segments(jgb)
        twice = jga[:sga[0][1]] + jgb[sgb[0][0]:sgb[0][1]] + jga[sga[0][1]:]
        oerr, opx, _ = oracle.decode(twice)
        assert oerr == 0 and np.array_equal(opx, oracle.decode(jgb)[1])
        outs, st = hip.decode([twice])
        assert st == [0] and np.array_equal(outs[0], opx)
        outs, st = hip.decode([jga, twice, jgb])
        assert st[0] == 0 and st[2] == 0 and st[1] == 0x1001
        assert np.array_equal(outs[0], oracle.decode(jga)[1]) and np.array_equal(outs[2], oracle.decode(jgb)[1])


def test_pipelined_stream_with_batches_that_are_decoded_again(hip, oracle):
    """mjpeg.decode_stream: three decoders, each with a stream on which its upload and its kernels run
    (jpeg_hip_decoder_set_upload_stream), the frames' states coming back in page-locked memory at the end of run().
    Batches that make sync() decode again go through the same machinery: restart markers the header does not
    announce (segment tables grown), a stray marker in the data (cut out on the host copy), data that ends inside an
    MCU; and a damaged frame leaves its neighbours alone.  Pageable frames and frames in one page-locked buffer."""
    from jpeg_amd import mjpeg
    frames = []
    for i in range(14):
        img = synth_image(208, 112, 3, 8, seed=500 + i, frame=i)
        err, j, _ = oracle.encode(img, P=8, H=2, V=1, q=70 + i, optimize=0, restart_interval=(0, 1, 4)[i % 3])
        assert err == 0
        if i % 3 == 1:
            dri = j.index(b"\xff\xdd")
            j = j[:dri + 4] + b"\x7f\xff" + j[dri + 6:]      # DRI promises next to no markers: far more come
        if i == 5:
            j = j[:len(j) * 3 // 5] + b"\xff\xd9"               # the data runs out
        if i == 9:
            at = _ecs_mid(j)
            j = j[:at] + b"\xff\xe3\x00\x04ab" + j[at:]         # an APP3 segment in the middle of the data
        frames.append(j)
    want = [oracle.decode(j) for j in frames]
    pinned = hip.PinnedFrames.from_frames(frames)
    for src in (frames, pinned, frames):
        k = 0
        for out, st in mjpeg.decode_stream(src, batch=3):
            px = out.cpu().numpy()
            for f in range(px.shape[0]):
                assert st[f] & 0xFFFFFFFF == want[k][0] & 0xFFFFFFFF, (k, hex(st[f]))
                if want[k][0] == 0:
                    assert np.array_equal(px[f].reshape(want[k][1].shape), want[k][1]), k
                k += 1
        assert k == len(frames)
    pinned.buf.close()


def _ecs_mid(jpg):
    """a place in the middle of the entropy-coded data that is not inside a stuffed byte or a marker"""
    sos = jpg.index(b"\xff\xda")
    at = sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3]) + (len(jpg) - sos) // 2
    while jpg[at - 1] == 0xFF or jpg[at] == 0xFF:
        at += 1
    return at


def test_upload_on_a_stream_of_the_callers(hip, oracle):
    """jpeg_hip_decoder_set_upload_stream: the upload queued on the caller's stream, run() on the same stream or on
    another one (it waits for the upload by event either way), then back to the decoder's own copy stream; one
    decoder object through all of it, batches of different sizes."""
    import torch
    imgs = [synth_image(144, 80, 3, 8, seed=700 + i, frame=i) for i in range(7)]
    jpgs = [oracle.encode(im, P=8, H=2, V=2, q=65 + 4 * i, optimize=i % 2, restart_interval=(0, 3)[i % 2])[1] for i, im in enumerate(imgs)]
    want = [oracle.decode(j)[1] for j in jpgs]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d = hip.Decoder()
    try:
        for up, run_on, part in ((s1, s1, jpgs[:5]), (s1, s2, jpgs[2:]), (None, s2, jpgs[:3]), (s2, s2, jpgs)):
            d.set_upload_stream(up)
            d.prepare(part)
            with torch.cuda.stream(run_on):
                out = d.run()
            assert d.sync() == [0] * len(part)
            k0 = jpgs.index(part[0])
            for k, px in enumerate(d.to_arrays(out)):
                assert np.array_equal(px, want[k0 + k])
    finally:
        d.close()


def test_decode_mixed_takes_every_golden_in_one_call(hip, oracle):
    """binding.decode_mixed(): files of any mix of sizes, sampling and scan layouts in one call -- grouped into batches
    by their headers, frames whose scans differ from their batch's decoded again on their own."""
    import json
    with open(golden_path("manifest.json")) as f:
        man = json.load(f)
    names = sorted(set(g["jpg_file"] for k in ("decode", "encode") for g in man[k].values() if "jpg_file" in g))
    jpgs = []
    for n in names:
        with open(golden_path(n), "rb") as fh:
            jpgs.append(fh.read())
    jpgs = jpgs + jpgs[::-1]                                   # every file twice, far apart
    px, st = hip.decode_mixed(jpgs)
    assert len(px) == len(st) == len(jpgs)
    for i, j in enumerate(jpgs):
        oerr, opx, _ = oracle.decode(j)
        assert (st[i] == 0) == (oerr == 0), (i, hex(st[i] & 0xffffffff), hex(oerr & 0xffffffff))
        if oerr == 0:
            assert np.array_equal(px[i], opx), i
