"""The stream pipelines must not depend on the process's history and must hand their
output tensors over to the consumer's stream properly."""
import time

import numpy as np
import pytest

from jpeg_amd.synth import synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from jpeg_amd import binding
    assert binding.lib().jpeg_hip_device_count() > 0, "no HIP device: the product path has no CPU fallback"
    return binding


def _side_by_side(torch, a, b, cycles=30_000_000):
    def run(x, y):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(x):
            torch.cuda._sleep(cycles)
        with torch.cuda.stream(y):
            torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    return run(a, b) < 0.75 * run(a, a)


@pytest.mark.parametrize("before", [0, 3, 5])
def test_stream_set_is_on_queues_of_its_own_whatever_came_before(hip, before):
    """HIP maps streams to its four hardware queues A B C D D C B A ... as they are made: three streams taken in a
    row from position 3 are D D C.  jpeg_hip_streams_create measures instead."""
    import torch
    torch.cuda.set_device(0)
    earlier = [torch.cuda.Stream(0) for _ in range(before)]          # the process's history
    for s in earlier:
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    ss = hip.StreamSet(0, 3)
    try:
        assert ss.proven == 3
        cur = torch.cuda.current_stream(0)
        for i in range(3):
            assert _side_by_side(torch, cur, ss.streams[i]), "stream %d shares a queue with the caller's stream" % i
            for j in range(i + 1, 3):
                assert _side_by_side(torch, ss.streams[i], ss.streams[j]), "streams %d and %d share a queue" % (i, j)
    finally:
        ss.close()


def test_decode_stream_hands_its_tensors_to_a_slow_consumer(hip, oracle):
    """Batches are yielded to a consumer that keeps the device busy on ITS stream for a long time before it looks at
    the tensor, while the pipeline queues the next batches: the tensor's memory must not be handed to a later run()
    meanwhile (record_stream), so every batch must still check out."""
    import torch
    from jpeg_amd import mjpeg
    torch.cuda.set_device(0)
    imgs = [synth_image(256, 128, 3, 8, seed=40 + i) for i in range(6)]
    jpgs, st = hip.encode(imgs, P=8, H=2, V=1, q=85, optimize=0)
    assert st == [0] * 6
    want = [oracle.decode(j)[1] for j in jpgs]
    frames = [jpgs[i % 6] for i in range(60)]
    sums = []
    for out, st in mjpeg.decode_stream(frames, batch=6, device=0):
        assert all(s == 0 for s in st)
        torch.cuda._sleep(20_000_000)                    # the consumer's stream is busy; the use below is queued behind it
        sums.append(out.to(torch.int64).sum(dim=1))      # `out` is dropped right after this line is QUEUED
        del out
    torch.cuda.synchronize()
    expect = [int(np.asarray(w, dtype=np.int64).sum()) for w in want]
    for b, s in enumerate(sums):
        assert [int(v) for v in s.tolist()] == expect, "batch %d was overwritten under the consumer" % b
    mjpeg.release_stream_decoders()
