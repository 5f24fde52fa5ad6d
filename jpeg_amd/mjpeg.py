"""Motion JPEG on one or several GPUs.

Every frame of a Motion JPEG stream is an independent SOI..EOI codestream -- the reference's
process_jpeg_stream() handles exactly one (decoder.c:603-612) -- so frames shard over GPUs with no
collective on the data path: one process per GPU, rank r of N takes the r-th contiguous chunk of
the frame list, decodes (or encodes) it in batches on its own device and keeps the results there.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to gather a few integers per
rank: frames, bytes, an order-independent checksum.

    frames = split(stream_bytes)                 # C++ splitter in libjpeg_amd.so (host only)
    lo, hi = shard(len(frames), rank, world)     # this rank's frames
    for out, status in decode_batches(frames[lo:hi], batch=256): ...
    totals = gather_stats({"frames": hi - lo, "bytes": nbytes, "checksum": cs}, dist)
"""
import ctypes as C
import zlib

from . import binding


def split(stream):
    """Frames of a concatenated JPEG stream as a list of bytes objects (order preserved)."""
    stream = bytes(stream)
    L = binding.lib()
    n = L.jpeg_hip_mjpeg_split(stream, len(stream), None, None, 0)
    if n < 0:
        raise binding.JpegError(n, "jpeg_hip_mjpeg_split")
    if n == 0:
        return []
    off = (C.c_size_t * n)()
    ln = (C.c_size_t * n)()
    m = L.jpeg_hip_mjpeg_split(stream, len(stream), off, ln, n)
    assert m == n
    return [stream[off[i]:off[i] + ln[i]] for i in range(n)]


def shard(nframes, rank, world):
    """Contiguous chunk [lo, hi) of rank `rank`: sizes differ by at most one frame, order preserved
    so that concatenating the ranks' outputs in rank order gives the single-GPU output."""
    base, extra = divmod(nframes, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def frame_checksum(data):
    return zlib.crc32(bytes(data)) & 0xffffffff


def combine_checksums(checksums):
    """Order-independent checksum of checksums (sum modulo 2^61 - 1 of a mixed value)."""
    acc = 0
    for c in checksums:
        acc = (acc + (c * 0x9e3779b97f4a7c15 + 0x7f4a7c15) % ((1 << 61) - 1)) % ((1 << 61) - 1)
    return acc


def decode_batches(frames, batch=256, device=0):
    """Decode `frames` (same geometry) in batches on `device`; yields (uint8 tensor [n, frame_bytes]
    resident on the device, per-frame status list).  One decoder object serves every batch: its device
    buffers only grow, so batches after the first allocate nothing."""
    d = binding.Decoder(device)
    try:
        for i in range(0, len(frames), batch):
            d.prepare(frames[i:i + batch])
            out = d.run()
            st = d.sync()
            yield out, st
    finally:
        d.close()


def decode_stream(frames, batch=256, device=0):
    """The same, pipelined: STREAM_DEPTH decoder objects take turns, each on a stream of
    its own.  A helper thread parses the next batches' headers and queues their upload on that decoder's stream (a
    DMA from the library's pinned staging buffer, or straight from the caller's memory when `frames` is a
    binding.PinnedFrames) while the kernels of this batch run on another stream; and the kernels of batch i+1 are
    queued BEFORE the caller waits for batch i, so the device does not idle while Python looks at statuses, and
    the narrow kernels at the end of one batch's synchronisation run beside the wide ones of the next.  No copy
    streams: a process has four hardware queues.  Yields (tensor, status) in order; the tensor of a batch stays
    valid until the batch after the next one has been yielded."""
    from concurrent.futures import ThreadPoolExecutor
    torch = binding._torch()
    decs, streams = _stream_decoders(device)
    n = len(decs)
    chunks = [frames[i:i + batch] for i in range(0, len(frames), batch)]
    if not chunks:
        return
    with ThreadPoolExecutor(max_workers=1) as pool:
        futs = {}                                  # batch -> its prepare() in the helper thread
        outs = {}                                  # batch -> its output tensor, queued
        try:
            for j in range(min(n, len(chunks))):
                futs[j] = pool.submit(decs[j % n].prepare, chunks[j])
            for i in range(len(chunks)):
                # batch i + 1 is queued before batch i is waited for (its prepare() is awaited if need be), and so
                # are the batches behind it, up to n - 1 ahead, if they happen to be prepared already
                for j in range(i, min(i + n, len(chunks))):
                    if j not in outs and j in futs and (j <= i + 1 or futs[j].done()):
                        futs.pop(j).result()
                        with torch.cuda.stream(streams[j % n]):
                            outs[j] = decs[j % n].run()
                st = decs[i % n].sync()            # waits for batch i only (an event behind its last kernel)
                if i + n < len(chunks):
                    futs[i + n] = pool.submit(decs[i % n].prepare, chunks[i + n])   # that decoder is free again
                yield outs.pop(i), st
        finally:
            for fut in futs.values():
                try:
                    fut.result()                   # a consumer that stops early: let the helper finish its batches
                except binding.JpegError:
                    pass


_stream_state = {}
STREAM_DEPTH = int(__import__("os").environ.get("JPEG_AMD_STREAM_DEPTH", "3"))   # decoders (and streams) that take turns
# 0 = normal.  A process that has had many streams busy before may find these three on ONE hardware queue (HIP maps
# streams to its four queues as they are first used) and the pipeline serialised: 46 k instead of 72 k frames/s
# measured; -1 (high priority: another set of queues) brought that case back to 67 - 73 k.  DESIGN.md section 5.
STREAM_PRIORITY = int(__import__("os").environ.get("JPEG_AMD_STREAM_PRIORITY", "0"))


def _stream_decoders(device):
    """The decoder objects and their streams of decode_stream(), kept between calls: their pinned staging
    buffers and device workspaces only grow, and making them anew costs more than decoding a few thousand
    frames (hipHostMalloc of 150 MB pins every page).  release_stream_decoders() frees them."""
    if device not in _stream_state:
        torch = binding._torch()
        if not _stream_state:
            import atexit
            atexit.register(release_stream_decoders)   # while the HIP runtime is still there
        decs = [binding.Decoder(device) for _ in range(STREAM_DEPTH)]
        streams = [torch.cuda.Stream(device=device, priority=STREAM_PRIORITY) for _ in range(STREAM_DEPTH)]
        for d, st in zip(decs, streams):
            d.set_upload_stream(st)
        _stream_state[device] = (decs, streams)
    return _stream_state[device]


def release_stream_decoders():
    for decs, _ in _stream_state.values():
        for d in decs:
            d.close()
    _stream_state.clear()


def gather_stats(local, dist=None):
    """Sum `frames` and `bytes`, combine `checksum` over all ranks.  dist = torch.distributed module
    with an initialised process group, or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {"frames": local["frames"], "bytes": local["bytes"], "checksum": local["checksum"],
                "per_rank": [dict(local)]}
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([local["frames"], local["bytes"], local["checksum"]], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    rows = [[int(x) for x in t.tolist()] for t in allv]
    m = (1 << 61) - 1
    return {"frames": sum(r[0] for r in rows), "bytes": sum(r[1] for r in rows),
            "checksum": sum(r[2] for r in rows) % m,
            "per_rank": [{"frames": r[0], "bytes": r[1], "checksum": r[2]} for r in rows]}
