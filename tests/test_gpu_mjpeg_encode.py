"""Motion JPEG stream out (the encode direction): a batch's codestreams leave the device packed
back to back, in page-locked host memory or in device memory, with a table of offsets.  The bytes of every frame
must be the oracle's codestream of that image (produce_codestream, encoder.c:589-645), and the packed run must read
back as a Motion JPEG stream (jpeg_hip_mjpeg_split)."""
import numpy as np
import pytest

from jpeg_amd.synth import synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _images(n, X, Y, nf=3, P=8):
    return [synth_image(X, Y, nf, P, seed=100 + i) for i in range(n)]


@pytest.mark.parametrize("optimize", [0, 1])
@pytest.mark.parametrize("where", ["pinned", "device"])
def test_packed_stream_equals_the_oracles_frames(hip, oracle, optimize, where):
    import torch
    from jpeg_amd import mjpeg
    X, Y = 203, 117                                     # unaligned: odd codestream lengths, every byte phase
    imgs = _images(7, X, Y)
    want = []
    for im in imgs:
        err, jpg, _ = oracle.encode(im, P=8, H=2, V=1, q=85, optimize=optimize)
        assert err == 0
        want.append(jpg)
    e = hip.Encoder()
    e.configure(len(imgs), X, Y, 3, P=8, H=2, V=1, q=85, optimize=optimize)
    samples = e.upload(imgs)
    room = sum(len(j) for j in want) + 64
    if where == "pinned":
        out = hip.PinnedBuffer(room)
        view = np.frombuffer(out.view(), dtype=np.uint8)
        view[:] = 0xAA
    else:
        out = torch.full((room,), 0xAA, dtype=torch.uint8, device="cuda:0")
    e.run_stream(samples, out)
    st, off = e.sync_stream()
    assert st == [0] * len(imgs)
    assert off[0] == 0 and off[-1] == sum(len(j) for j in want)
    got = bytes(view) if where == "pinned" else bytes(out.cpu().numpy())
    for f, jpg in enumerate(want):
        assert got[off[f]:off[f + 1]] == jpg, "frame %d differs from the oracle's codestream" % f
        assert e.jpeg(f, len(jpg)) == jpg                # the per-frame copy still works after a stream run
    assert got[off[-1]:] == b"\xaa" * (room - off[-1])   # nothing written behind the stream
    assert mjpeg.split(got[:off[-1]]) == want
    # a plain run() afterwards is not a stream run
    e.run(samples)
    assert e.sync()[0] == [0] * len(imgs)
    with pytest.raises(hip.JpegError):
        e.sync_stream()
    e.close()


def test_frames_that_fail_or_do_not_fit(hip, oracle):
    # 12-bit: frame 1 leaves the DC range (the reference assert()s, encoder.c:447) and takes no room
    good = synth_image(64, 16, 1, 12, seed=3, range_safe=True)
    bad = np.zeros((16, 64), np.uint16)
    bad[:, 32:] = 4095
    imgs = [good, bad, good[::-1].copy()]
    e = hip.Encoder()
    e.configure(3, 64, 16, 1, P=12, H=1, V=1, q=100, optimize=1)
    out = hip.PinnedBuffer(1 << 16)
    e.run_stream(e.upload(imgs), out)
    st, off = e.sync_stream()
    assert st[0] == 0 and st[1] == hip.JPEG_HIP_ERR_DC_RANGE and st[2] == 0
    assert off[1] == off[2] and off[3] > off[2]
    view = bytes(np.frombuffer(out.view(), dtype=np.uint8)[:off[-1]])
    for f in (0, 2):
        err, jpg, _ = oracle.encode(imgs[f], P=12, q=100, optimize=1)
        assert err == 0 and view[off[f]:off[f + 1]] == jpg
    # room for two frames and a bit: the third is complete on the device and dropped from the stream, like all behind it
    imgs = _images(5, 96, 64)
    e.configure(5, 96, 64, 3, P=8, H=2, V=2, q=75, optimize=0)
    want = [oracle.encode(im, P=8, H=2, V=2, q=75, optimize=0)[1] for im in imgs]
    cap = len(want[0]) + len(want[1]) + len(want[2]) - 1
    e.run_stream(e.upload(imgs), out, cap=cap)
    st, off = e.sync_stream()
    assert st == [0, 0] + [hip.RET_FAILURE_OVERFLOW_ERROR] * 3
    assert off[2] == off[-1] == len(want[0]) + len(want[1])
    view = bytes(np.frombuffer(out.view(), dtype=np.uint8)[:off[-1]])
    assert view == want[0] + want[1]
    # memory the device cannot write is refused, not faulted on
    import ctypes as C
    plain = (C.c_ubyte * 4096)()
    err = hip.lib().jpeg_hip_encoder_run_stream(e.h, C.c_void_p(e.upload(imgs).data_ptr()), e.frame_bytes,
                                                C.cast(plain, C.c_void_p), 4096, None)
    assert err == hip.RET_FAILURE_LOGIC_ERROR
    e.close()
    out.close()


def test_encode_stream_pipeline(hip, oracle):
    """mjpeg.encode_stream: three encoders deep, batches of different sizes, the views of earlier batches stay valid
    while later ones are in flight."""
    import torch
    from jpeg_amd import mjpeg
    X, Y = 160, 96
    imgs = _images(11, X, Y)
    want = [oracle.encode(im, P=8, H=2, V=1, q=85, optimize=0)[1] for im in imgs]
    e0 = hip.Encoder()
    e0.configure(4, X, Y, 3, P=8, H=2, V=1, q=85, optimize=0)
    dev = e0.upload(imgs)                               # [11, frame_bytes]
    batches = [dev[0:4], dev[4:8], dev[8:11]]
    got = []
    for view, off, st in mjpeg.encode_stream(iter(batches), X, Y, 3, P=8, H=2, V=1, q=85, optimize=0):
        assert all(s == 0 for s in st)
        got.extend(bytes(view[off[f]:off[f + 1]]) for f in range(len(st)))
    assert got == want
    # and again (the encoders are kept), consumed lazily
    it = mjpeg.encode_stream(iter(batches), X, Y, 3, P=8, H=2, V=1, q=85, optimize=0)
    view, off, st = next(it)
    assert mjpeg.split(bytes(view)) == want[:4]
    assert [bytes(v[o[f]:o[f + 1]]) for v, o, s in it for f in range(len(s))] == want[4:]
    e0.close()
