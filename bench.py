#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (children of this
process through torch.distributed.run, before anything here touches a GPU) and exits with their code.

Workload (config.workload): BASELINE.json's metric configuration, 4K (3840x2160) 8-bit 4:2:0 q90,
default Huffman tables, no restart markers -- the stream the reference encoder itself writes.  One
STEP = one pass of the hot path over one batch of F DISTINCT synthetic frames per GPU: the batch of
codestreams is DECODED (un-stuffing, speculative Huffman decode, IDCT, colour) and the batch of
images is ENCODED (colour, FDCT, quantise, Huffman bit-packing, stuffing).  Inputs are resident in
HBM before the timed region; outputs stay in HBM.  value = pixels that went through decode AND
encode per second, whole job (all ranks).  Frames shard across ranks with no data-path
collective (Motion-JPEG semantics) => "scaling": "weak".

Extra keys: decode / encode rates measured separately; "workloads": the other BASELINE configurations
(C2 4K 4:4:4 q90 with one restart interval per MCU row, C3 8K 12-bit gray, C4 8K 4:2:0 optimised
tables, C5 1080p 4:2:2 Motion JPEG), each with its own algorithmic bytes and dominant kernel;
"mjpeg_stream": BASELINE's 10 000-frame 1080p stream sharded over the ranks, host-inclusive
(codestreams in host memory -> samples in HBM), per rank and in total, with the frame count and a
checksum of checksums gathered over the ranks and compared with the single-process totals;
"roofline" for the dominant kernel (HIP-event time per launch against algorithmic bytes, SURVEY.md
section 8d) and "cpu_baseline" (the compiled reference, oracle/_ref, on one host core).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
STREAM_FRAMES = 10000          # BASELINE.json configs[4]: Motion JPEG 1080p 4:2:2 q85, 10k frames


HOST_SYNTH = 0                 # --host-synth N: N distinct frames from the numpy generator, repeated (rocprofv3 --pmc runs)


def make_batch(torch, binding, synth, name, nframes, device, restart_interval=0):
    """nframes DISTINCT synthetic frames, generated on the device (jpeg_amd.synth.synth_body_torch: the
    integer-only recipe of SURVEY 8d, bit-identical to the numpy generator the parity tests use)
    -> (encoder, device tensor of PPM/PGM bodies, JPEG codestreams from OUR encoder (bit-exact vs the
    oracle, tests/), configuration)."""
    X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS[name]
    dev = "cuda:%d" % device
    if HOST_SYNTH:
        body = lambda a: a.astype(np.uint8).tobytes() if P <= 8 else a.astype(">u2").tobytes()
        base = [torch.frombuffer(bytearray(body(synth.synth_image(X, Y, nc, P, seed, safe, frame=i))), dtype=torch.uint8)
                for i in range(min(HOST_SYNTH, nframes))]
        samples = torch.stack([base[i % len(base)] for i in range(nframes)]).to(dev)
    else:
        samples = torch.stack([synth.synth_body_torch(X, Y, nc, P, seed, safe, frame=i, device=dev) for i in range(nframes)])
    enc = binding.Encoder(device)
    enc.configure(nframes, X, Y, nc, P, H, V, q, opt, restart_interval)
    enc.run(samples)
    st, ln = enc.sync()
    assert all(s == 0 for s in st), st
    jpgs = [enc.jpeg(f, ln[f]) for f in range(nframes)]
    return enc, samples, jpgs, dict(X=X, Y=Y, nc=nc, P=P, H=H, V=V, q=q, opt=opt, Ri=restart_interval)


def host_image(samples, f, cfg):
    """frame f of a device batch as the numpy array the CPU codecs take"""
    raw = samples[f].cpu().numpy().tobytes()
    a = np.frombuffer(raw, np.uint8 if cfg["P"] <= 8 else np.dtype(">u2"))
    a = a.reshape(cfg["Y"], cfg["X"], cfg["nc"]).astype(np.uint8 if cfg["P"] <= 8 else np.uint16)
    return a[:, :, 0] if cfg["nc"] == 1 else a


def timed(torch, fn, steps, warmup, dist):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t = time.perf_counter() - t0
    if dist is not None:
        t = reduce_max(torch, dist, t)
    return t


def reduce_max(torch, dist, v):
    tt = torch.tensor([v], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def kernel_times(obj, run, torch, reps=5):
    """Average per-kernel device time (ms) over reps launches, hipEvents on the launch stream."""
    obj.set_timing(True)
    acc = {}
    for _ in range(reps):
        run()
        torch.cuda.synchronize()
        for k, v in obj.timing().items():
            acc[k] = acc.get(k, 0.0) + v
    obj.set_timing(False)
    return {k: v / reps for k, v in acc.items()}


def csrc_sha16():
    """what the profiles are tagged with: a hash of the kernel sources (the GPU box has no .git to ask)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "jpeg_amd", "csrc", "*"))):
        if path.endswith((".hip", ".hpp", ".inc", ".cpp")):
            h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


def newest_profile(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.basename(files[-1])


def profile_source(d, fname):
    """which committed profile a counter figure comes from, and whether the kernels have changed since (loudly)"""
    src = {"file": "profiles/" + fname, "head": d.get("head"), "csrc_sha16": d.get("csrc_sha16")}
    src["matches_this_tree"] = d.get("csrc_sha16") == csrc_sha16()
    if not src["matches_this_tree"]:
        sys.stderr.write("bench.py: WARNING: %s was collected on other kernel sources (csrc_sha16 %s, this tree %s): the "
                         "counter figures of the roofline block are STALE -- re-run tools/profile.sh\n"
                         % (src["file"], d.get("csrc_sha16"), csrc_sha16()))
    return src


def pmc_traffic(kernel, frames):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary
    (profiles/rNN_*_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950
    correction applied), scaled to this run's frames per launch, and where it came from.  (None, None) if
    there is no such file."""
    d, fname = newest_profile("pmc_traffic.json")
    k = d.get("kernels", {}).get(kernel) if d else None
    if not k:
        return None, None
    return int(k["hbm_bytes"] * frames / d["frames_per_launch"]), profile_source(d, fname)


def pmc_valu(kernel, frames, kernel_ms, clock_ghz=2.4, simds=1024):
    """The ceiling a VALU-bound kernel runs against: vector instructions per wave from the newest committed SQ
    counter passes, and the share of the launch's SIMD cycles in which the vector unit was busy with them
    (SQ_ACTIVE_INST_VALU counts units of four cycles, one per ordinary instruction; 256 CUs x 4 SIMDs at the peak
    engine clock), against this run's live kernel time."""
    d, fname = newest_profile("pmc_sq.json")
    k = d.get("kernels", {}).get(kernel) if d else None
    if not k or "SQ_ACTIVE_INST_VALU" not in k["counters"]:
        return None
    c = k["counters"]
    scale = frames / d["frames_per_launch"]
    busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] * scale / (simds * clock_ghz * 1e9 * kernel_ms * 1e-3)
    return {"insts_per_wave": k["derived"].get("valu_insts_per_wave"), "waves_per_launch": int(c.get("SQ_WAVES", 0) * scale),
            "busy_frac": round(busy, 4), "clock_ghz": clock_ghz, "source": profile_source(d, fname)}


def measure_workload(torch, binding, synth, name, nframes, device, steps, restart_interval=0, lanes=1):
    """decode and encode rates of one BASELINE configuration on one GPU, device-resident, with the per-kernel
    HIP-event times and the algorithmic-byte roofline of its dominant kernel."""
    enc, samples, jpgs, cfg = make_batch(torch, binding, synth, name, nframes, device, restart_interval)
    dec = binding.Decoder(device)
    dec.prepare(jpgs)
    out = dec.alloc_output()
    pix = cfg["X"] * cfg["Y"]
    t_d1 = timed(torch, lambda: dec.run(out), steps, 2, None)
    assert all(s == 0 for s in dec.sync())
    sums1 = frame_sums(torch, out)
    lanes_got = dec.set_lanes(lanes) if lanes > 1 else 1
    out.zero_()
    t_d = timed(torch, lambda: dec.run(out), steps, 2, None) if lanes > 1 else t_d1
    t_e = timed(torch, lambda: enc.run(samples), steps, 2, None)
    assert all(s == 0 for s in dec.sync()) and all(s == 0 for s in enc.sync()[0])
    assert torch.equal(frame_sums(torch, out), sums1), "the lanes decoded something else than one stream"
    stats = dec.stats(0)
    allk = kernel_times(dec, lambda: dec.run(out), torch, reps=3)
    allk.update(kernel_times(enc, lambda: enc.run(samples), torch, reps=3))
    dom = max(allk, key=allk.get)
    sbytes = 1 if cfg["P"] <= 8 else 2
    alg = nframes * pix * cfg["nc"] * sbytes + sum(len(j) for j in jpgs)
    res = {
        "frames_per_step": nframes, "distinct_frames": nframes,
        "decode_mpix_s": round(nframes * pix * steps / t_d / 1e6, 1),
        "decode_mpix_s_one_lane": round(nframes * pix * steps / t_d1 / 1e6, 1), "decoder_lanes": lanes_got,
        "encode_mpix_s": round(nframes * pix * steps / t_e / 1e6, 1),
        "jpeg_bytes_per_frame": sum(len(j) for j in jpgs) // nframes,
        "alg_bytes_per_step": alg,
        "pipeline_hbm_frac": {"decode": round(alg / (t_d / steps) / 1e9 / HBM_PEAK_GBS, 5),
                              "encode": round(alg / (t_e / steps) / 1e9 / HBM_PEAK_GBS, 5)},
        "dominant_kernel": dom, "dominant_kernel_ms": round(allk[dom], 4),
        "dominant_kernel_frac": round(alg / (allk[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "kernel_ms": {k: round(v, 4) for k, v in sorted(allk.items(), key=lambda kv: -kv[1])[:6]},
        "huffman_serial_fallback": bool(stats[0]), "restart_segments_per_frame": stats[3],
    }
    enc.close()
    dec.close()
    return res


def frame_sums(torch, out):
    """per-frame checksum of a decoded batch, computed on the device: the frame's bytes summed as 64-bit words
    (wrapping), 61 bits kept"""
    n = out.shape[1]
    if n % 8 == 0 and out.is_contiguous():
        s = out.view(torch.int64).sum(dim=1)
    else:
        s = out.sum(dim=1, dtype=torch.int64)
    return s & ((1 << 61) - 1)


def mjpeg_stream(torch, binding, synth, mjpeg, dev, rank, world, dist, batch, distinct=32):
    """BASELINE configs[4]: a 10 000-frame 1080p 4:2:2 q85 Motion JPEG stream sharded over the ranks with
    mjpeg.shard (contiguous chunks, no data-path collective).  Host-inclusive: codestreams in host memory ->
    decoded samples in HBM (header parsing, Huffman table set-up, PCIe upload, kernels)."""
    enc, samples, jpgs, cfg = make_batch(torch, binding, synth, "c5_1080p_422_q85", distinct, dev)
    enc.close()
    del samples
    stream = [jpgs[i % distinct] for i in range(STREAM_FRAMES)]
    lo, hi = mjpeg.shard(STREAM_FRAMES, rank, world)
    mine = stream[lo:hi]
    shard_bytes = sum(len(j) for j in mine)

    # expected totals of the WHOLE stream as one process would see them: the stream repeats `distinct` frames
    d = binding.Decoder(dev)
    d.prepare(jpgs)
    o = d.run()
    assert all(s == 0 for s in d.sync())
    sums = [int(v) for v in frame_sums(torch, o).tolist()]
    d.close()
    del o
    expect_cs = mjpeg.combine_checksums(sums[i % distinct] for i in range(STREAM_FRAMES))
    expect_bytes = sum(len(stream[i]) for i in range(STREAM_FRAMES))

    def timed_pass(src):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        n = 0
        acc = []
        for o, st in mjpeg.decode_stream(src, batch=batch, device=dev):
            assert all(s == 0 for s in st)
            acc.append(frame_sums(torch, o))
            n += o.shape[0]
        torch.cuda.synchronize()
        t_rank = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        t_all = time.perf_counter() - t0
        if dist is not None:
            t_all = reduce_max(torch, dist, t_all)
        # EVERY pass is checked, the reported one included: frames, bytes and the checksum of checksums, gathered
        # over the ranks, against what one process decoding the whole stream sees (outside the timed region)
        got = [int(v) for v in torch.cat(acc).tolist()]
        tot = mjpeg.gather_stats({"frames": n, "bytes": shard_bytes, "checksum": mjpeg.combine_checksums(got)}, dist)
        assert tot["frames"] == STREAM_FRAMES and tot["bytes"] == expect_bytes and tot["checksum"] == expect_cs, tot
        return n, tot, t_rank, t_all

    # (a) the rank's part of the file in ONE page-locked buffer (what an I/O layer that owns its buffers delivers:
    # every byte is read from DRAM by the DMA engine, no staging copy); (b) frames as pageable Python bytes objects
    # (the library copies them into its pinned staging buffer with a few threads, then DMA).  Three passes each,
    # taking turns: the first of each kind creates workspaces, the others show the run-to-run spread.
    pinned = binding.PinnedFrames.from_frames(mine)
    mine = [bytes(bytearray(j)) for j in mine]             # every frame its own pageable memory (not 32 cache-warm objects)
    runs = {"pinned": [], "pageable": []}
    for attempt in range(3):                               # interleaved: neither kind gets the warmer device
        for kind, src in (("pinned", pinned), ("pageable", mine)):
            n, tot, t_rank, t_all = timed_pass(src)
            runs[kind].append((t_rank, t_all))
    pinned.buf.close()
    best = min(runs["pinned"][1:], key=lambda r: r[1])
    ok = True                                              # every pass asserted its totals above
    rates = [n / best[0]]
    if dist is not None:
        tt = torch.tensor([n / best[0]], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        rates = [float(v.item()) for v in allv]
    return {
        "frames": tot["frames"], "distinct_frames": distinct, "batch": batch, "bytes_per_frame": expect_bytes // STREAM_FRAMES,
        "host_inclusive_decode_frames_per_s": round(STREAM_FRAMES / best[1], 1),
        "input": "each rank's frames back to back in one page-locked host buffer (jpeg_hip_decoder_prepare_pinned)",
        "passes_frames_per_s": {k: [round(STREAM_FRAMES / r[1], 1) for r in v] for k, v in runs.items()},
        "pageable_input_frames_per_s": round(STREAM_FRAMES / min(r[1] for r in runs["pageable"][1:]), 1),
        "per_rank_frames_per_s": [round(r, 1) for r in rates],
        "gathered": {"frames": tot["frames"], "bytes": tot["bytes"], "checksum": tot["checksum"]},
        "matches_single_process_totals": ok,
    }


def mjpeg_encode_stream(torch, binding, synth, mjpeg, dev, rank, world, dist, batch, distinct=32):
    """The encode direction of BASELINE configs[4]: 10 000 frames of 1080p 4:2:2 q85, sharded like the decode
    stream.  Host-inclusive on the OUTPUT side: samples resident in HBM (where a renderer, a capture DMA or the
    decoder leaves them; uploading 6.2 MB per frame would cap the stream at the link's 8.8 k frames/s) -> finished
    codestreams packed back to back in page-locked host memory, ready to be written out as a Motion JPEG file."""
    X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS["c5_1080p_422_q85"]
    enc, samples, jpgs, cfg = make_batch(torch, binding, synth, "c5_1080p_422_q85", distinct, dev)
    enc.close()
    lo, hi = mjpeg.shard(STREAM_FRAMES, rank, world)
    big = torch.cat([samples] * ((batch + distinct - 1) // distinct))[:batch].contiguous()    # one batch, frame i = image i % distinct
    del samples
    assert batch % distinct == 0 or world * batch >= STREAM_FRAMES
    want_len = [len(jpgs[i % distinct]) for i in range(batch)]
    counts = [min(batch, hi - lo - k) for k in range(0, hi - lo, batch)]
    expect_bytes = 0                                       # of the whole stream: every rank's batches start at image 0
    for r in range(world):
        rlo, rhi = mjpeg.shard(STREAM_FRAMES, r, world)
        expect_bytes += sum(sum(want_len[:min(batch, rhi - rlo - k)]) for k in range(0, rhi - rlo, batch))

    def batches():
        for n in counts:
            yield big[:n]

    def one_pass(check_bytes):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        frames = nbytes = 0
        lens_ok = True
        for view, off, st in mjpeg.encode_stream(batches(), X, Y, nc, P=P, H=H, V=V, q=q, optimize=opt, device=dev):
            n = len(st)
            lens_ok = lens_ok and not any(st) and [off[f + 1] - off[f] for f in range(n)] == want_len[:n]
            if check_bytes:                                # the untimed pass: every byte of every frame
                for f in range(n):
                    assert bytes(view[off[f]:off[f + 1]]) == jpgs[f % distinct], "frame %d of a batch differs" % f
            frames += n
            nbytes += off[-1]
        t_rank = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        t_all = time.perf_counter() - t0
        if dist is not None:
            t_all = reduce_max(torch, dist, t_all)
        assert lens_ok
        tot = mjpeg.gather_stats({"frames": frames, "bytes": nbytes, "checksum": 0}, dist)
        assert tot["frames"] == STREAM_FRAMES and tot["bytes"] == expect_bytes, tot
        return frames, tot, t_rank, t_all

    one_pass(True)                                         # creates the workspaces; compares every byte with the codestreams of
                                                           # the encoder object (which tests/ pin to the oracle)
    runs = [one_pass(False) for _ in range(3)]
    best = min(runs, key=lambda r: r[3])
    rates = [best[0] / best[2]]
    if dist is not None:
        tt = torch.tensor([rates[0]], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        allv = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allv, tt)
        rates = [float(v.item()) for v in allv]
    return {
        "frames": best[1]["frames"], "distinct_frames": distinct, "batch": batch, "bytes_per_frame": best[1]["bytes"] // STREAM_FRAMES,
        "host_inclusive_encode_frames_per_s": round(STREAM_FRAMES / best[3], 1),
        "output": "each batch's codestreams back to back in page-locked host memory (jpeg_hip_encoder_run_stream); samples resident in HBM",
        "passes_frames_per_s": [round(STREAM_FRAMES / r[3], 1) for r in runs],
        "per_rank_frames_per_s": [round(r, 1) for r in rates],
        "verified": "status, frame count and every frame's length in every pass; every byte in an untimed pass",
    }


def cpu_baseline(imgs, jpgs, cfg, budget_s=12.0):
    """The compiled reference (oracle/_ref: unmodified sources) on ONE host core: decode + encode of
    whole 4K frames, in process, files in memory/tmpfs.  Bounded sample (~10-30 s)."""
    try:
        from oracle.pyoracle import Reference
        ref = Reference()
        kind = "reference"
        enc = lambda im: ref.encode(im, P=cfg["P"], H=cfg["H"], V=cfg["V"], q=cfg["q"], optimize=cfg["opt"])
        dec = lambda j: ref.decode(j)
    except Exception as e:
        sys.stderr.write("bench.py: WARNING: oracle/_ref (the compiled reference) is not usable here (%r); the CPU "
                         "baseline below times OUR scalar restatement instead -> cpu_baseline.kind = \"port\"\n" % (e,))
        from oracle.pyoracle import Oracle
        orc = Oracle()
        kind = "port"
        enc = lambda im: orc.encode(im, P=cfg["P"], H=cfg["H"], V=cfg["V"], q=cfg["q"], optimize=cfg["opt"])
        dec = lambda j: orc.decode(j)
    pix = cfg["X"] * cfg["Y"]
    n = 0
    t_dec = t_enc = 0.0
    t_start = time.perf_counter()
    while n == 0 or time.perf_counter() - t_start < budget_s:
        t0 = time.perf_counter()
        e, _, _ = dec(jpgs[n % len(jpgs)])
        t1 = time.perf_counter()
        e2, _, _ = enc(imgs[n % len(imgs)])
        t2 = time.perf_counter()
        assert e == 0 and e2 == 0
        t_dec += t1 - t0
        t_enc += t2 - t1
        n += 1
    return {
        "value": round(n * pix / (t_dec + t_enc) / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": kind,
        "sample": "%d x (decode + encode) of a 4K 4:2:0 q90 frame, in-process, %.1f s" % (n, t_dec + t_enc),
        "decode_mpix_s": round(n * pix / t_dec / 1e6, 3), "encode_mpix_s": round(n * pix / t_enc / 1e6, 3),
    }


def cpu_worker(seconds):
    """One host core of the Motion-JPEG CPU baseline (started by cpu_baseline_mjpeg as a child process; no
    GPU, no torch): the compiled reference decodes and encodes 1080p 4:2:2 q85 frames until the time is up."""
    from jpeg_amd import synth
    from oracle.pyoracle import Reference
    ref = Reference()
    X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS["c5_1080p_422_q85"]
    imgs = [synth.synth_image(X, Y, nc, P, seed, safe, frame=i) for i in range(2)]
    jpgs = [ref.encode(im, P=P, H=H, V=V, q=q, optimize=opt)[1] for im in imgs]
    n = 0
    t_dec = t_enc = 0.0
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < seconds:
        t0 = time.perf_counter()
        e = ref.decode(jpgs[n % 2])[0]
        t1 = time.perf_counter()
        e2 = ref.encode(imgs[n % 2], P=P, H=H, V=V, q=q, optimize=opt)[0]
        t2 = time.perf_counter()
        assert e == 0 and e2 == 0
        t_dec += t1 - t0
        t_enc += t2 - t1
        n += 1
    print(json.dumps({"frames": n, "t_dec": t_dec, "t_enc": t_enc}))


def cpu_baseline_mjpeg(seconds=8.0):
    """frames are independent, so the CPU side of the Motion-JPEG comparison is N copies of the
    single-threaded reference on N host cores (N = this box's CPU share, at most 16).  Children of this
    process, each a plain python without GPU."""
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT) for _ in range(cores)]
    dec = enc = 0.0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 6 + 60)
        r = json.loads(out.decode().strip().splitlines()[-1])
        dec += r["frames"] / r["t_dec"]
        enc += r["frames"] / r["t_enc"]
    return {"decode_frames_per_s": round(dec, 1), "encode_frames_per_s": round(enc, 1), "cores": cores,
            "sample": "%d processes x %.0f s of decode + encode of 1080p 4:2:2 q85 frames, compiled reference" % (cores, seconds)}


def spawn_ranks(n):
    """`python bench.py --gpus N` on its own: start the N ranks as children (torch.distributed.run, one process
    per GPU) BEFORE this process touches a GPU, pass their output through, exit with their code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128, help="4K frames per GPU per step")
    ap.add_argument("--mjpeg-frames", type=int, default=256, help="1080p frames per GPU per batch (Motion JPEG lines)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mjpeg", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the C2..C5 lines")
    ap.add_argument("--host-synth", type=int, default=0,
                    help="make only this many distinct frames, on the host, and repeat them (rocprofv3 --pmc runs: the "
                         "counter collection of ROCm 7.2 crashes inside torch's int64 kernels of the device-side generator)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo rehearses the multi-rank path "
                                                      "with several ranks on ONE GPU (RCCL refuses that)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="streams the decoder lays a batch out over (jpeg_hip_decoder_set_lanes); with the encoder's "
                         "stream beside them, 3 fills the four hardware queues of a process")
    ap.add_argument("--one-stream", action="store_true",
                    help="decode and encode on ONE stream (for rocprofv3 --stats: kernels of the two pipelines then "
                         "do not overlap and their durations are the ones `kernel_ms` reports)")
    args = ap.parse_args()
    if args.cpu_worker > 0:
        cpu_worker(args.cpu_worker)
        return
    global HOST_SYNTH
    HOST_SYNTH = args.host_synth
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher decides, n_gpus = %d\n" % (args.gpus, world, world))
    dist = None
    if args.backend != "nccl":
        local = local % torch.cuda.device_count()          # rehearsal: ranks share the GPUs there are
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist_mod.init_process_group(args.backend, rank=rank, world_size=world)
        dist = dist_mod
    else:
        torch.cuda.set_device(local)
    from jpeg_amd import binding, synth, mjpeg

    dev = local
    F = args.frames
    enc, samples, jpgs, cfg = make_batch(torch, binding, synth, "metric_4k_420_q90", F, dev)
    dec = binding.Decoder(dev)
    dec.prepare(jpgs)
    out = dec.alloc_output()
    pix = cfg["X"] * cfg["Y"]

    # the decode batch and the encode batch of a step are independent jobs: each runs on a stream of its
    # own, so the short, narrow kernels of one pipeline (scans, walk rounds) overlap the wide ones of the other
    # (streams picked by measurement, a hardware queue each: the decoder's own stream, the encoder's, and the
    # decoder's other lanes -- jpeg_hip_decoder_set_lanes: narrow kernels of one run of frames beside the wide ones
    # of the others)
    lanes = 1 if args.one_stream else max(1, min(args.lanes, 4))
    if args.one_stream:
        s_dec = s_enc = torch.cuda.Stream(dev)
        queues = None
    else:
        queues = binding.StreamSet(dev, min(4, lanes + 1), avoid_current=False)
        s_dec, s_enc = queues.streams[0], queues.streams[1]
        lane_streams = queues.streams[2:2 + lanes - 1]
        while len(lane_streams) < lanes - 1:             # more lanes than queues: the last ones share the encoder's
            lane_streams.append(s_enc)
        with torch.cuda.stream(s_dec):
            dec.set_lanes(lanes, lane_streams)

    def step():
        with torch.cuda.stream(s_dec):
            dec.run(out)
        with torch.cuda.stream(s_enc):
            enc.run(samples)

    # correctness guard inside the bench: the decoded batch must be what the encoder was fed (lossy bound)
    dec.run(out)
    st = dec.sync()
    assert all(s == 0 for s in st), st
    for f in (0, F - 1):
        diff = (out[f].to(torch.int16) - samples[f].to(torch.int16)).abs().max().item()
        assert diff < 64, diff

    t = timed(torch, step, args.steps, args.warmup, dist)
    assert all(s == 0 for s in dec.sync()) and all(s == 0 for s in enc.sync()[0])
    value = world * F * pix * args.steps / t / 1e6

    def dec_alone():
        with torch.cuda.stream(s_dec):
            dec.run(out)

    def enc_alone():
        with torch.cuda.stream(s_enc):
            enc.run(samples)

    t_d = timed(torch, dec_alone, args.steps, 1, dist)
    t_e = timed(torch, enc_alone, args.steps, 1, dist)
    t_d1 = t_d
    if lanes > 1:
        sums = frame_sums(torch, out)
        dec.set_lanes(1)
        out.zero_()
        t_d1 = timed(torch, dec_alone, args.steps, 1, dist)
        assert torch.equal(frame_sums(torch, out), sums), "the lanes decoded something else than one stream"

    res = {
        "metric": "Mpix/s decode+encode, 4K 4:2:0 q90 (every pixel is decoded AND encoded once per step)",
        "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "dtype_detail": "pixel kernels: binary32 DCT / quantisation / RGB->YCbCr (8-bit), binary64 colour maths elsewhere, bit-exact with the reference; "
                        "entropy kernels: u8 / int16 / u32 integer work",
        "config": {"workload": "metric_4k_420_q90: 3840x2160 8-bit 4:2:0 q90, default Huffman tables, no restart markers",
                   "frames_per_gpu_per_step": F, "distinct_frames": min(F, HOST_SYNTH) if HOST_SYNTH else F,
                   "parallelism": "frames sharded over %d GPU(s), no collectives" % world},
        "decode_mpix_s": round(world * F * pix * args.steps / t_d / 1e6, 1),
        "decode_mpix_s_one_lane": round(world * F * pix * args.steps / t_d1 / 1e6, 1), "decoder_lanes": lanes,
        "encode_mpix_s": round(world * F * pix * args.steps / t_e / 1e6, 1),
    }

    if rank == 0 and world == 1 and not args.no_workloads:
        # The decoder's synchronisation ends in a few narrow kernels whose time does not shrink with the batch (the
        # slowest chains decide it): a second decoder on twice the frames shows what they cost at this batch size.
        dec2 = binding.Decoder(dev)
        dec2.prepare(jpgs + jpgs)
        if lanes > 1:
            dec2.set_lanes(lanes)
        out2 = dec2.alloc_output()
        dec2.run(out2)
        assert all(s == 0 for s in dec2.sync())
        n2 = max(3, args.steps // 2)
        t2 = timed(torch, lambda: dec2.run(out2), n2, 1, None)
        res["decode_mpix_s_at_%d_frames" % (2 * F)] = round(2 * F * pix * n2 / t2 / 1e6, 1)
        dec2.close()
        del out2
    if rank == 0:
        # ---- roofline of the dominant kernel, live HIP-event timing on the launch stream ----
        kd = kernel_times(dec, lambda: dec.run(out), torch)
        ke = kernel_times(enc, lambda: enc.run(samples), torch)
        allk = dict(kd)
        allk.update(ke)
        dom = max(allk, key=allk.get)
        jpg_bytes = sum(len(j) for j in jpgs)
        alg_bytes = F * pix * cfg["nc"] + jpg_bytes        # W*H*Nc*s + B_jpeg per frame
        achieved = alg_bytes / (allk[dom] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom, F)
        res["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "traffic_source": traffic_src, "valu": pmc_valu(dom, F, allk[dom]),
                           "alg_bytes_per_launch": alg_bytes, "kernel_ms": round(allk[dom], 4)}
        res["kernel_ms"] = {k: round(v, 4) for k, v in sorted(allk.items(), key=lambda kv: -kv[1])}
        res["pipeline_hbm_frac"] = {"decode": round(alg_bytes / (t_d / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                                    "encode": round(alg_bytes / (t_e / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}
    if rank == 0 and world == 1 and not args.no_workloads:
        # ---- the reference's own use: ONE image per call (decoder in.jpg out.ppm / encoder in.ppm out.jpg).  Host
        # inclusive on the side that is small: codestream from host memory -> samples in HBM (prepare + run + sync);
        # samples in HBM -> codestream in host memory (run + sync + copy).  Objects kept between calls.
        d1 = binding.Decoder(dev)
        e1 = binding.Encoder(dev)
        e1.configure(1, cfg["X"], cfg["Y"], cfg["nc"], cfg["P"], cfg["H"], cfg["V"], cfg["q"], cfg["opt"])
        one = samples[:1]

        def dec_one():
            d1.prepare(jpgs[:1])
            o1 = d1.run()
            assert d1.sync() == [0]
            return o1

        def enc_one():
            e1.run(one)
            st1, ln1 = e1.sync()
            assert st1 == [0]
            return e1.jpeg(0, ln1[0])

        dec_one()
        assert enc_one() == jpgs[0]
        reps = 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dec_one()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            enc_one()
        t2 = time.perf_counter()
        res["single_frame_ms"] = {"decode_jpg_in_host_memory_to_samples_in_hbm": round((t1 - t0) / reps * 1e3, 3),
                                  "encode_samples_in_hbm_to_jpg_in_host_memory": round((t2 - t1) / reps * 1e3, 3),
                                  "frame": "4K 4:2:0 q90, one frame per call, objects kept between calls"}
        d1.close()
        e1.close()
    cpu_imgs = [host_image(samples, f, cfg) for f in range(4)] if rank == 0 and world == 1 and not args.no_cpu else None
    cpu_jpgs = jpgs[:4]
    enc.close()
    dec.close()
    if queues is not None:
        queues.close()
    del out, samples

    if rank == 0 and world == 1 and not args.no_workloads:
        # ---- the other BASELINE configurations, one GPU, device-resident -----------------------
        res["workloads"] = {}
        for name, nf, ri in (("c2_4k_444_q90", 32, 480), ("c3_8k_gray12_q95", 16, 0), ("c4_8k_420_q95", 16, 0),
                             ("c5_1080p_422_q85", args.mjpeg_frames, 0)):
            # (no encoder beside the decoder in these lines: a fourth lane takes the queue the encoder has in `value`)
            w = measure_workload(torch, binding, synth, name, nf, dev, args.steps, ri, lanes + 1 if lanes > 1 else 1)
            if name.startswith("c5"):
                w["decode_frames_per_s"] = round(w["decode_mpix_s"] * 1e6 / (1920 * 1080), 1)
                w["encode_frames_per_s"] = round(w["encode_mpix_s"] * 1e6 / (1920 * 1080), 1)
            res["workloads"][name + ("_ri%d" % ri if ri else "")] = w

    if not args.no_mjpeg:
        ms = mjpeg_stream(torch, binding, synth, mjpeg, dev, rank, world, dist, args.mjpeg_frames)
        assert ms["matches_single_process_totals"], ms
        res["mjpeg_stream_1080p_422_q85"] = ms
        mjpeg.release_stream_decoders()
        es = mjpeg_encode_stream(torch, binding, synth, mjpeg, dev, rank, world, dist, args.mjpeg_frames)
        res["mjpeg_stream_1080p_422_q85"].update({
            "host_inclusive_encode_frames_per_s": es["host_inclusive_encode_frames_per_s"], "encode": es})
        mjpeg.release_stream_decoders()

    if rank == 0 and world == 1 and not args.no_cpu:
        res["cpu_baseline"] = cpu_baseline(cpu_imgs, cpu_jpgs, cfg)
        if not args.no_mjpeg:
            try:
                res["cpu_baseline"]["mjpeg_1080p_422_q85"] = cpu_baseline_mjpeg()
            except Exception as e:                         # the headline line must not depend on this extra
                res["cpu_baseline"]["mjpeg_1080p_422_q85"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
