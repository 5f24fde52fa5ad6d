#!/usr/bin/env python3
"""Developer tool: which of torch's pool streams share a hardware queue (torch.cuda._sleep on two streams: side by
side or one after the other?).  Usage: python tools/queue_probe_torch.py [streams] [priority]"""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
streams = [torch.cuda.Stream(0, priority=prio) for _ in range(n)]
CY = 40_000_000


def pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(CY)
    with torch.cuda.stream(b):
        torch.cuda._sleep(CY)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for s in streams:
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()
one = pair(streams[0], streams[0]) / 2
print("one sleep: %.1f ms; ids %s" % (one * 1e3, [hex(s.cuda_stream)[-6:] for s in streams]))
for i in range(n):
    row = []
    for j in range(n):
        row.append("  ." if j <= i else ("  S" if pair(streams[i], streams[j]) > 1.5 * one else "  -"))
    print("%3d:" % i + "".join(row))
d = torch.cuda.default_stream(0)
print("def:" + "".join("  S" if pair(d, streams[j]) > 1.5 * one else "  -" for j in range(n)))
