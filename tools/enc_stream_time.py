#!/usr/bin/env python3
"""Developer tool: time of the stream-out kernels into page-locked host memory and into device memory."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth
import bench

torch.cuda.set_device(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, "c5_1080p_422_q85", F, 0)
room = sum(len(j) for j in jpgs) + 4096
pin = binding.PinnedBuffer(room)
dev = torch.empty(room, dtype=torch.uint8, device="cuda:0")
for name, out in (("pinned", pin), ("device", dev)):
    enc.set_timing(True)
    for _ in range(3):
        enc.run_stream(samples, out)
        enc.sync_stream()
    print(name, {k: round(v, 3) for k, v in enc.timing().items()})
    enc.set_timing(False)
    t0 = time.perf_counter()
    for _ in range(10):
        enc.run_stream(samples, out)
        enc.sync_stream()
    t = (time.perf_counter() - t0) / 10
    print(name, "ms per batch %.3f -> %.1f frames/s, %.1f GB/s out" % (t * 1e3, F / t, room / t / 1e9))
# DMA for comparison: device -> pinned, exact size
torch.cuda.synchronize()
host = torch.empty(room, dtype=torch.uint8).pin_memory()
t0 = time.perf_counter()
for _ in range(10):
    host.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
print("hipMemcpy D2H of %d MB: %.3f ms, %.1f GB/s" % (room >> 20, t * 1e3, room / t / 1e9))
