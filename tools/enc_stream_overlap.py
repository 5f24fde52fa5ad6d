#!/usr/bin/env python3
"""Developer tool: do the stream-out batches of several encoders overlap?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth
import bench

torch.cuda.set_device(0)
F = 256
enc0, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, "c5_1080p_422_q85", F, 0)
room = sum(len(j) for j in jpgs) + 4096
X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS["c5_1080p_422_q85"]
N = 3
encs = [enc0] + [binding.Encoder(0) for _ in range(N - 1)]
for e in encs[1:]:
    e.configure(F, X, Y, nc, P, H, V, q, opt)
ss = binding.StreamSet(0, N)
print("proven", ss.proven)
for kind in ("device", "pinned"):
    outs = [binding.PinnedBuffer(room) if kind == "pinned" else torch.empty(room, dtype=torch.uint8, device="cuda:0") for _ in range(N)]
    for n in (1, 2, 3):
        def rnd():
            for i in range(n):
                with torch.cuda.stream(ss.streams[i]):
                    encs[i].run_stream(samples, outs[i])
            for i in range(n):
                encs[i].sync_stream()
        rnd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            rnd()
        t = (time.perf_counter() - t0) / 10
        print(kind, n, "encoders: %.3f ms per round, %.1f frames/s" % (t * 1e3, n * F / t))
    # plain run() on n streams for comparison
    for n in (1, 3):
        def rnd2():
            for i in range(n):
                with torch.cuda.stream(ss.streams[i]):
                    encs[i].run(samples)
            for i in range(n):
                encs[i].sync()
        rnd2()
        t0 = time.perf_counter()
        for _ in range(10):
            rnd2()
        t = (time.perf_counter() - t0) / 10
        print("plain run", n, "encoders: %.3f ms per round, %.1f frames/s" % (t * 1e3, n * F / t))
