"""Motion JPEG host layer on CPU: frame splitting (C++ splitter through the C-ABI, no device
needed), sharding, and the world_size-2 statistics gather over gloo.  The oracle is the checker."""
import os
import socket
import struct

import numpy as np
import pytest

from jpeg_amd.synth import synth_image
from tests.conftest import golden_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(oracle, n=7, strip_dht=False):
    frames = []
    for i in range(n):
        img = synth_image(96, 64, 3, 8, seed=300 + i, frame=i)
        err, jpg, _ = oracle.encode(img, P=8, H=2, V=1, q=85, optimize=0)
        assert err == 0
        frames.append(jpg)
    return frames


def test_split_finds_every_frame(oracle):
    from jpeg_amd import mjpeg
    frames = _stream(oracle)
    assert mjpeg.split(b"".join(frames)) == frames
    # garbage between frames is skipped, order is kept
    noisy = b"\x00\x01junk" + frames[0] + b"\xff\xff\x00garbage" + frames[1] + b"tail"
    assert mjpeg.split(noisy) == frames[:2]
    assert mjpeg.split(b"") == [] and mjpeg.split(b"no jpeg here") == []


def test_split_is_not_fooled_by_marker_bytes_in_payloads(oracle):
    from jpeg_amd import mjpeg
    frames = _stream(oracle, 2)
    # an APP1 segment whose payload contains FF D9 and FF D8
    payload = b"Exif\x00\x00\xff\xd9\xff\xd8\xff\xda"
    app1 = b"\xff\xe1" + struct.pack(">H", len(payload) + 2) + payload
    tricky = frames[0][:2] + app1 + frames[0][2:]
    got = mjpeg.split(tricky + frames[1])
    assert got == [tricky, frames[1]]
    assert oracle.decode(tricky)[0] == 0
    # a frame without EOI runs to the end of the data
    assert mjpeg.split(frames[0] + frames[1][:-2]) == [frames[0], frames[1][:-2]]


def test_split_on_reference_style_files():
    from jpeg_amd import mjpeg
    with open(golden_path("d_two_frames.jpg"), "rb") as f:
        two = f.read()
    got = mjpeg.split(two)
    assert len(got) == 2 and b"".join(got) == two
    with open(golden_path("d_pil_444_rst_row.jpg"), "rb") as f:
        rst = f.read()                                  # restart markers inside the scan are not frame ends
    assert mjpeg.split(rst + rst) == [rst, rst]


def test_shard_covers_everything_once():
    from jpeg_amd import mjpeg
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            spans = [mjpeg.shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, stream, q):
    import torch.distributed as dist
    from jpeg_amd import mjpeg
    from oracle.pyoracle import Oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        orc = Oracle()
        frames = mjpeg.split(stream)
        lo, hi = mjpeg.shard(len(frames), rank, world)
        sums, nbytes = [], 0
        for j in frames[lo:hi]:
            err, px, _ = orc.decode(j)                   # the checker stands in for the device on CPU
            assert err == 0
            sums.append(mjpeg.frame_checksum(px.tobytes()))
            nbytes += px.nbytes
        tot = mjpeg.gather_stats({"frames": hi - lo, "bytes": nbytes, "checksum": mjpeg.combine_checksums(sums)}, dist)
        if rank == 0:
            q.put(tot)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo_agree_with_one_process(oracle):
    """world_size 2 on CPU: each rank takes its shard, totals and the checksum of checksums equal the
    single-process run (no data-path collective: only three integers per rank are gathered)."""
    import torch.multiprocessing as mp
    from jpeg_amd import mjpeg
    frames = _stream(oracle, 7)
    stream = b"".join(frames)
    sums, nbytes = [], 0
    for j in frames:
        px = oracle.decode(j)[1]
        sums.append(mjpeg.frame_checksum(px.tobytes()))
        nbytes += px.nbytes
    want = mjpeg.combine_checksums(sums)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, stream, q)) for r in range(2)]
    for p in procs:
        p.start()
    tot = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tot["frames"] == 7 and tot["bytes"] == nbytes and tot["checksum"] == want
    assert [r["frames"] for r in tot["per_rank"]] == [4, 3]
