#!/usr/bin/env python3
"""Developer tool: the Motion JPEG decode pipeline from pinned memory, alone and beside a busy host thread."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jpeg_amd import binding, synth, mjpeg
import bench
torch.cuda.set_device(0)
enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, "c5_1080p_422_q85", 32, 0)
enc.close()
frames = [jpgs[i % 32] for i in range(10000)]
pin = binding.PinnedFrames.from_frames(frames)
mine = [bytes(bytearray(j)) for j in frames]

def one(src):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for o, st in mjpeg.decode_stream(src, batch=256, device=0):
        n += o.shape[0]
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)

print("pinned  ", [round(one(pin)) for _ in range(3)])
print("pageable", [round(one(mine)) for _ in range(3)])
stop = False
def spin():
    a = np.ones(1 << 16)
    while not stop:
        a.sum()
ths = [threading.Thread(target=spin) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4)]
for t in ths: t.start()
print("pinned + busy host threads  ", [round(one(pin)) for _ in range(3)])
print("pageable + busy host threads", [round(one(mine)) for _ in range(3)])
stop = True
for t in ths: t.join()
