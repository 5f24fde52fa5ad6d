"""jpeg_hip_decoder_set_lanes(): a batch laid out over several streams of one decoder object decodes to the same
samples, coefficients and statuses as on one stream -- and as the oracle says."""
import numpy as np
import pytest

from jpeg_amd.synth import synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _batch(hip, n, X, Y, nc, P, H, V, q, opt, ri):
    imgs = [synth_image(X, Y, nc, P, seed=300 + i) for i in range(n)]
    jpgs, st = hip.encode(imgs, P=P, H=H, V=V, q=q, optimize=opt, restart_interval=ri)
    assert st == [0] * n
    return jpgs


@pytest.mark.parametrize("lanes", [2, 3, 4])
@pytest.mark.parametrize("case", [
    (13, 208, 120, 3, 8, 2, 2, 90, 0, 0),
    (7, 333, 77, 3, 8, 2, 1, 75, 1, 3),
    (5, 640, 256, 1, 12, 1, 1, 75, 0, 0),
    (2, 96, 64, 3, 8, 1, 1, 50, 1, 0),          # fewer frames than lanes (for 3 and 4)
])
def test_lanes_decode_what_one_stream_decodes(hip, oracle, lanes, case):
    import torch
    torch.cuda.set_device(0)
    n = case[0]
    jpgs = _batch(hip, *case)
    want = [oracle.decode(j) for j in jpgs]
    d = hip.Decoder()
    try:
        assert 1 <= d.set_lanes(lanes) <= lanes
        for rep in range(2):                             # the second run() reuses streams and events
            d.prepare(jpgs)
            out = d.run()
            assert d.sync() == [0] * n
            for f, px in enumerate(d.to_arrays(out)):
                oerr, opx, oinfo = want[f]
                assert oerr == 0 and np.array_equal(px, opx), (lanes, f, rep)
        d.set_lanes(1)                                   # and back
        out = d.run()
        assert d.sync() == [0] * n
        for f, px in enumerate(d.to_arrays(out)):
            assert np.array_equal(px, want[f][1])
    finally:
        d.close()


def test_lanes_keep_stream_order_and_frame_statuses(hip, oracle):
    """Work queued on the caller's stream before run() is seen by every lane, work queued after run() sees every
    lane's result; a damaged frame in the middle keeps its status and its neighbours their samples."""
    import torch
    torch.cuda.set_device(0)
    jpgs = _batch(hip, 9, 256, 144, 3, 8, 2, 2, 85, 0, 0)
    bad = bytearray(jpgs[4])
    bad[len(bad) // 2] ^= 0x5a
    cut = jpgs[6][:len(jpgs[6]) // 3]
    mixed = jpgs[:4] + [bytes(bad)] + [jpgs[5]] + [cut] + jpgs[7:]
    want = [oracle.decode(j) for j in mixed]
    one = hip.Decoder()
    one.prepare(mixed)
    ref_out = one.run()
    ref_st = one.sync()
    ref = ref_out.clone()
    one.close()
    d = hip.Decoder()
    try:
        side = torch.cuda.Stream(0)
        with torch.cuda.stream(side):
            assert d.set_lanes(3) >= 1
            d.prepare(mixed)
            out = d.alloc_output()
            out.fill_(0x77)                              # queued before run(): no lane may start ahead of it
            torch.cuda._sleep(20_000_000)
            d.run(out)
            got = out.clone()                            # queued behind run(): every lane must have finished
        side.synchronize()
        st = d.sync()
        assert st == ref_st
        assert torch.equal(got, ref)
        for f in range(9):
            oerr, opx, _ = want[f]
            if oerr == 0 and st[f] == 0:
                assert np.array_equal(d.to_arrays(got)[f], opx), f
    finally:
        d.close()


def test_lanes_from_the_environment(hip, oracle, monkeypatch):
    import torch
    torch.cuda.set_device(0)
    monkeypatch.setenv("JPEG_AMD_LANES", "3")
    jpgs = _batch(hip, 6, 160, 96, 3, 8, 2, 2, 80, 0, 0)
    px, st = hip.decode(jpgs)
    assert st == [0] * 6
    for f in range(6):
        assert np.array_equal(px[f], oracle.decode(jpgs[f])[1])
