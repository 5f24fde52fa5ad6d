#!/usr/bin/env python3
"""Developer tool: per-kernel times of a 1080p batch prepared from pageable memory (frames 4 KiB-aligned on the
device) and from one pinned buffer (frames back to back as they lie in the file)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth
import bench
torch.cuda.set_device(0)
F = 256
enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, "c5_1080p_422_q85", 32, 0)
frames = [jpgs[i % 32] for i in range(F)]
pin = binding.PinnedFrames.from_frames(frames)
for name, src in (("pageable", frames), ("pinned", pin), ("pageable", frames), ("pinned", pin)):
    d = binding.Decoder(0)
    t0 = time.perf_counter()
    for _ in range(5):
        d.prepare(src)
        torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / 5
    out = d.alloc_output()
    d.run(out); d.sync()
    k = bench.kernel_times(d, lambda: d.run(out), torch, 5)
    t = bench.timed(torch, lambda: d.run(out), 10, 2, None) / 10
    print(name, "prepare+upload %.3f ms, run %.3f ms" % (tp * 1e3, t * 1e3), {a: round(b, 3) for a, b in sorted(k.items(), key=lambda kv: -kv[1])[:7]})
    d.close()
