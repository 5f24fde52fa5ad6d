#!/usr/bin/env python3
"""Experiment helper: time k_encode_pixels / k_decode_pixels of a library VARIANT (JPEG_AMD_LIB=path),
results not checked."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth
if os.environ.get("JPEG_AMD_LIB"):
    binding.LIB_PATH = os.environ["JPEG_AMD_LIB"]
F = 64
X, Y, nc, P, H, V, q, opt, safe, seed = synth.CONFIGS["metric_4k_420_q90"]
imgs = [synth.synth_image(X, Y, nc, P, seed, safe, frame=i) for i in range(4)]
enc = binding.Encoder(0)
enc.configure(F, X, Y, nc, P, H, V, q, opt)
samples = enc.upload([imgs[i % 4] for i in range(F)])
enc.set_timing(True)
acc = {}
for it in range(8):
    enc.run(samples)
    torch.cuda.synchronize()
    for k, v in enc.timing().items():
        if it >= 3:
            acc[k] = acc.get(k, 0) + v / 5
res = {k: round(v, 4) for k, v in acc.items() if "pixels" in k}
st, ln = enc.sync()
jpgs = [enc.jpeg(f, ln[f]) for f in range(F)]
dec = binding.Decoder(0)
dec.prepare(jpgs)
out = dec.alloc_output()
dec.set_timing(True)
acc = {}
for it in range(8):
    dec.run(out)
    torch.cuda.synchronize()
    for k, v in dec.timing().items():
        if it >= 3:
            acc[k] = acc.get(k, 0) + v / 5
res.update({k: round(v, 4) for k, v in acc.items() if "pixels" in k})
print(os.environ.get("JPEG_AMD_LIB", "default"), json.dumps(res))
