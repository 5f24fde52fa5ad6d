#!/usr/bin/env python3
"""Developer tool: H2D rate from page-locked memory, one small region over and over against a walk through a large one."""
import time
import torch
torch.cuda.set_device(0)
piece = 124 << 20
big = torch.empty(38 * piece, dtype=torch.uint8).pin_memory()
big.fill_(1)
dst = torch.empty(piece, dtype=torch.uint8, device="cuda:0")
for name, offs in (("same 124 MB", [0] * 38), ("walk 4.6 GB", [i * piece for i in range(38)]), ("same 124 MB", [0] * 38),
                   ("walk 4.6 GB", [i * piece for i in range(38)])):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o in offs:
        dst.copy_(big[o:o + piece], non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%s: %.1f GB/s" % (name, len(offs) * piece / t / 1e9))
