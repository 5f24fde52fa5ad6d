#!/usr/bin/env python3
"""Debug helper: decode one golden file on the GPU, compare coefficients with the CPU oracle block by block."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from jpeg_amd import binding
from oracle.pyoracle import Oracle

name = sys.argv[1]
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
jpg = open(os.path.join("tests/golden", name), "rb").read()
err, dec, info = Oracle().decode(jpg)
d = binding.Decoder()
d.prepare([jpg])
if dbg:
    d.set_debug(dbg)
out = d.run()
print("status", d.sync(), "stats", d.stats(0))
for k, c in enumerate(info.comps):
    a = d.coefficients(0)[k]
    bad = np.nonzero((a != c["coef"]).any(axis=1))[0]
    print("comp", k, "blocks", len(a), "bad", len(bad), bad[:40].tolist())
    for b in bad[:6]:
        print("   blk", b, "hip", a[b][:6].tolist(), "orc", c["coef"][b][:6].tolist())
