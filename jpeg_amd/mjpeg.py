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
    streams: a process has four hardware queues.  Yields (tensor, status) in order.  sync() has waited for the batch,
    so the consumer may use the tensor on its own (the current) stream at once; the tensor is the consumer's to keep or
    drop -- its memory is handed back to the allocator only after the consumer's work queued so far has run
    (record_stream)."""
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
                out = outs.pop(i)
                # The tensor was allocated on the decoder's stream and is about to be used on the consumer's: tell the
                # caching allocator, or the block goes back to that stream's pool the moment the consumer drops the
                # tensor and the NEXT run() queued there may overwrite it under a consumer kernel still pending.
                out.record_stream(torch.cuda.current_stream(device))
                yield out, st
        finally:
            for fut in futs.values():
                try:
                    fut.result()                   # a consumer that stops early: let the helper finish its batches
                except binding.JpegError:
                    pass


def encode_stream(batches, X, Y, Nf, P=8, H=2, V=1, q=75, optimize=0, restart_interval=0, device=0, out_bytes=None):
    """The encode direction of the stream layer (the reference writes its one codestream as
    it goes, encoder.c:589-645): `batches` yields uint8 device tensors [n, frame_bytes] of PPM/PGM bodies (n may
    vary up to the first batch's); every batch leaves the device as ONE run of bytes -- its codestreams back to
    back in page-locked host memory, written by the encoder's last kernels across the link -- and is yielded as
    (numpy uint8 view of that memory, frame_off[n + 1], per-frame status): frame f is view[off[f]:off[f + 1]], the
    view is a Motion JPEG stream as split() reads it.  STREAM_DEPTH encoder objects take turns, a stream and a host
    buffer each: batch i + 1 (and i + 2) is queued before batch i is waited for, so the stores of one batch cross
    the link beside the kernels of the next.  The view of a batch stays valid until STREAM_DEPTH - 1 more batches
    have been yielded.  out_bytes: room per batch in host memory (default: a quarter of the samples' bytes, at least
    1 MB; a batch that needs more comes back with RET_FAILURE_OVERFLOW_ERROR for the frames that did not fit)."""
    import numpy as np
    torch = binding._torch()
    encs, streams, bufs = _stream_encoders(device)
    n = len(encs)
    it = iter(batches)
    pending = []                                   # (batch index, frames) queued and not yet yielded
    cfg = {}

    def queue(j, samples):
        e = encs[j % n]
        key = (int(samples.shape[0]), X, Y, Nf, P, H, V, q, optimize, restart_interval)
        if cfg.get(j % n) != key:
            e.configure(samples.shape[0], X, Y, Nf, P=P, H=H, V=V, q=q, optimize=optimize, restart_interval=restart_interval)
            cfg[j % n] = key
        want = out_bytes or max(1 << 20, samples.numel() // 4)
        if bufs[j % n] is None or bufs[j % n].nbytes < want:
            if bufs[j % n] is not None:
                bufs[j % n].close()
            bufs[j % n] = binding.PinnedBuffer(want)
        st = streams[j % n]
        st.wait_stream(torch.cuda.current_stream(device))     # the samples were made on the caller's stream
        samples.record_stream(st)
        with torch.cuda.stream(st):
            e.run_stream(samples, bufs[j % n])
        pending.append(j)

    j = 0
    for samples in it:
        queue(j, samples)
        j += 1
        if len(pending) == n:                      # n - 1 batches stay in flight behind the one that is waited for
            i = pending.pop(0)
            st, off = encs[i % n].sync_stream()
            yield np.frombuffer(bufs[i % n].view(), dtype=np.uint8)[:off[-1]], off, st
    while pending:
        i = pending.pop(0)
        st, off = encs[i % n].sync_stream()
        yield np.frombuffer(bufs[i % n].view(), dtype=np.uint8)[:off[-1]], off, st


_enc_stream_state = {}


def _stream_encoders(device):
    """Encoder objects, streams and page-locked output buffers of encode_stream(), kept between calls."""
    if device not in _enc_stream_state:
        torch = binding._torch()
        if not _stream_state and not _enc_stream_state:
            import atexit
            atexit.register(release_stream_decoders)
        encs = [binding.Encoder(device) for _ in range(STREAM_DEPTH)]
        _enc_stream_state[device] = (encs, _pipeline_streams(device), [None] * STREAM_DEPTH)
    return _enc_stream_state[device]


_stream_state = {}
_stream_sets = {}
STREAM_DEPTH = int(__import__("os").environ.get("JPEG_AMD_STREAM_DEPTH", "3"))   # decoders / encoders (and streams) that take turns


def _pipeline_streams(device):
    """STREAM_DEPTH streams that were MEASURED to run side by side and beside the caller's stream
    (jpeg_hip_streams_create): HIP maps streams to its four hardware queues in an order that depends on how many the
    process made before, and three streams taken in a row can land two on one queue -- the pipeline then ran at 46 k
    instead of 72 k frames/s in a process that had used other streams first.  One set per
    device, shared by the decode and the encode pipeline."""
    if device not in _stream_sets:
        _stream_sets[device] = binding.StreamSet(device, min(4, STREAM_DEPTH))
    s = _stream_sets[device].streams
    return [s[i % len(s)] for i in range(STREAM_DEPTH)]


def _stream_decoders(device):
    """The decoder objects and their streams of decode_stream(), kept between calls: their pinned staging
    buffers and device workspaces only grow, and making them anew costs more than decoding a few thousand
    frames (hipHostMalloc of 150 MB pins every page).  release_stream_decoders() frees them."""
    if device not in _stream_state:
        torch = binding._torch()
        if not _stream_state and not _enc_stream_state:
            import atexit
            atexit.register(release_stream_decoders)   # while the HIP runtime is still there
        decs = [binding.Decoder(device) for _ in range(STREAM_DEPTH)]
        streams = _pipeline_streams(device)
        for d, st in zip(decs, streams):
            d.set_upload_stream(st)
            d.set_lanes(1)          # the pipeline keeps the queues busy with its decoders (whatever JPEG_AMD_LANES says:
                                    # two lanes per decoder were measured, 76 - 78 k frames/s either way)
        _stream_state[device] = (decs, streams)
    return _stream_state[device]


def release_stream_decoders():
    for decs, _ in _stream_state.values():
        for d in decs:
            d.close()
    _stream_state.clear()
    for encs, _, bufs in _enc_stream_state.values():
        for e in encs:
            e.close()
        for b in bufs:
            if b is not None:
                b.close()
    _enc_stream_state.clear()
    for ss in _stream_sets.values():
        ss.close()
    _stream_sets.clear()


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
