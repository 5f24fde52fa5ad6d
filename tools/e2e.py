#!/usr/bin/env python3
"""Host-inclusive timing of the Motion JPEG decode path (1080p 4:2:2 q85): everything from codestreams in
pageable host memory to samples on the device (and, optionally, back in pinned host memory).

    python tools/e2e.py [frames_per_batch] [batches]
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth, mjpeg
from bench import make_batch

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
enc, samples, jpgs, imgs, cfg = make_batch(binding, synth, "c5_1080p_422_q85", F, 0, distinct=4)
enc.close()
del samples
frames = jpgs * NB
host = [torch.empty((F, cfg["X"] * cfg["Y"] * 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
copy_stream = torch.cuda.Stream()
res = {"frames_per_batch": F, "batches": NB, "jpg_MB_per_batch": round(sum(map(len, jpgs)) / 1e6, 1)}

# phases of one batch, nothing overlapped
dec = binding.Decoder(0)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.prepare(jpgs)
    t1 = time.perf_counter()
    out = dec.run()
    dec.sync()
    t2 = time.perf_counter()
    host[0].copy_(out)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
res["one_batch_ms"] = {"prepare": round((t1 - t0) * 1e3, 2), "run_sync": round((t2 - t1) * 1e3, 2), "d2h": round((t3 - t2) * 1e3, 2)}
dec.close()

for name, fn in (("decode_batches", mjpeg.decode_batches), ("decode_stream", mjpeg.decode_stream)):
    for d2h in (False, True):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for i, (out, st) in enumerate(fn(frames, batch=F)):
                assert all(s == 0 for s in st)
                if d2h:
                    copy_stream.synchronize()              # the pinned buffer of two batches ago is free again
                    with torch.cuda.stream(copy_stream):
                        host[i & 1].copy_(out, non_blocking=True)
                n += out.shape[0]
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
        res["%s%s_fps" % (name, "_d2h" if d2h else "")] = round(n / t, 1)
print(json.dumps(res))
