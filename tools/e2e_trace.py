#!/usr/bin/env python3
"""Where the time of mjpeg.decode_stream goes: wall time of prepare() in the helper thread and of run()+sync()
in the caller, per batch, with and without the other side running.

    python tools/e2e_trace.py [batches] [frames per batch]
"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS["c5_1080p_422_q85"]
imgs = [synth.synth_image(X, Y, nc, P, seed, safe, frame=i) for i in range(4)]
jpgs, st = binding.encode(imgs, P=P, H=H, V=V, q=q, optimize=opt)
frames = [jpgs[i % 4] for i in range(F)]
decs = [binding.Decoder(0), binding.Decoder(0)]
stream = torch.cuda.Stream(device=0)

def T():
    return time.perf_counter()

# alone
for d in decs:
    d.prepare(frames)
t0 = T(); decs[0].prepare(frames); t1 = T()
with torch.cuda.stream(stream):
    out = decs[0].run()
decs[0].sync(); t2 = T()
with torch.cuda.stream(stream):
    out = decs[0].run()
decs[0].sync(); t3 = T()
print("alone: prepare %.2f ms, run+sync %.2f ms (second %.2f ms)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))

# overlapped, as decode_stream does
from concurrent.futures import ThreadPoolExecutor
log = []
def prep(d, i):
    a = T(); d.prepare(frames); b = T()
    log.append(("prepare", i, a, b))
with ThreadPoolExecutor(max_workers=1) as pool:
    start = T()
    fut = pool.submit(prep, decs[0], 0)
    for i in range(nb):
        a = T(); fut.result(); b = T()
        d = decs[i & 1]
        if i + 1 < nb:
            fut = pool.submit(prep, decs[(i + 1) & 1], i + 1)
        with torch.cuda.stream(stream):
            out = d.run()
        c = T()
        d.sync()
        e = T()
        log.append(("wait", i, a, b)); log.append(("launch", i, b, c)); log.append(("sync", i, c, e))
    total = T() - start
print("overlapped: %.2f ms per batch of %d = %.0f frames/s" % (total / nb * 1e3, F, nb * F / total))
for kind in ("prepare", "wait", "launch", "sync"):
    v = [(b - a) * 1e3 for k, i, a, b in log if k == kind and i >= 2]
    print("  %-8s mean %.2f ms  min %.2f  max %.2f" % (kind, sum(v) / len(v), min(v), max(v)))
