"""BASELINE.json's configurations at FULL size on the GPU, each against
the oracle on the same seeded input: codestream bytes, coefficients and samples bit-exact."""
import hashlib

import numpy as np
import pytest

from jpeg_amd.synth import CONFIGS, synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _run(hip, oracle, name, restart_interval=0, frames=1):
    X, Y, nc, P, H, V, q, opt, safe, seed = CONFIGS[name]
    imgs = [synth_image(X, Y, nc, P, seed, safe, frame=i) for i in range(frames)]
    # encode on the GPU: the reference encoder's stream (no restart markers), byte for byte
    jpgs, st = hip.encode(imgs, P=P, H=H, V=V, q=q, optimize=opt)
    assert st == [0] * frames
    err, ojpg, oinfo = oracle.encode(imgs[0], P=P, H=H, V=V, q=q, optimize=opt)
    assert err == 0
    assert hashlib.md5(jpgs[0]).hexdigest() == hashlib.md5(ojpg).hexdigest()
    # decode on the GPU (optionally the restart-marker variant of the same image, written by the oracle)
    src = jpgs
    if restart_interval:
        # the restart-marker variant: GPU encoder (extension) == oracle, byte for byte
        err, rj, _ = oracle.encode(imgs[0], P=P, H=H, V=V, q=q, optimize=opt, restart_interval=restart_interval)
        assert err == 0
        src, st = hip.encode(imgs, P=P, H=H, V=V, q=q, optimize=opt, restart_interval=restart_interval)
        assert st == [0] * frames and src[0] == rj
    d = hip.Decoder()
    d.prepare(src)
    out = d.run()
    assert d.sync() == [0] * frames
    err, opx, dinfo = oracle.decode(src[0])
    assert err == 0
    for a, c in zip(d.coefficients(0), dinfo.comps):
        assert np.array_equal(a, c["coef"])
    px = d.to_arrays(out)
    assert np.array_equal(px[0], opx)
    stats = d.stats(0)
    d.close()
    return px, stats


def test_c2_4k_444_q90_restart_per_mcu_row(hip, oracle):
    # 480 MCUs per row at 4:4:4: one restart interval per MCU row, encode AND decode (the encoder side of
    # restart markers is an extension of the reference; tests/test_gpu_parity.py has the reference decoder judge it)
    px, stats = _run(hip, oracle, "c2_4k_444_q90", restart_interval=480)
    assert stats[3] == 270                               # restart segments found


def test_c3_8k_gray12_q95_no_markers(hip, oracle):
    # one 27 MB entropy-coded segment, 12-bit samples: speculative decode over ~210 000 sub-sequences
    px, stats = _run(hip, oracle, "c3_8k_gray12_q95")
    assert px[0].dtype == np.uint16 and stats[0] == 0    # parallel path, no serial walk


def test_c4_8k_420_q95_optimised_tables(hip, oracle):
    _run(hip, oracle, "c4_8k_420_q95")


def test_c5_1080p_422_q85_batch(hip, oracle):
    # non-MCU-aligned height (1080 = 67.5 MCU rows): padding quirk Q7 included
    px, _ = _run(hip, oracle, "c5_1080p_422_q85", frames=6)
    assert len(px) == 6
