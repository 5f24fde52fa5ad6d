#!/usr/bin/env python3
"""Developer tool: re-run chosen cases of tools/fuzz_corrupt.py (same generator, same seed) and print the blocks
that differ.  python tools/dbg_corrupt.py <seed> <case> [<case> ...]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from jpeg_amd import binding
from jpeg_amd.synth import synth_image
from oracle.pyoracle import Oracle
from tools.fuzz_cases import corrupt_case

seed = int(sys.argv[1])
want = set(int(v) for v in sys.argv[2:])
rnd = random.Random(seed)
orc = Oracle()
for case in range(max(want) + 1):
    bad, meta = corrupt_case(rnd, orc, seed, case)
    if bad is None or case not in want:
        continue
    oerr, opx, oinfo = orc.decode(bad)
    d = binding.Decoder()
    d.prepare([bad])
    out = d.run()
    st = d.sync()[0]
    print("case", case, meta, "status", hex(st), hex(oerr), "mblocks", oinfo.mblocks if oinfo else None, "stats", d.stats(0))
    if st == 0 and oerr == 0:
        for k, (a, c) in enumerate(zip(d.coefficients(0), oinfo.comps)):
            for b in np.nonzero((a != c["coef"]).any(axis=1))[0][:6]:
                print("  comp", k, "block", b, "of", c["bx"], "x", c["by"])
                print("    gpu   ", a[b][:20].tolist(), "nz", int((a[b] != 0).sum()))
                print("    oracle", c["coef"][b][:20].tolist(), "nz", int((c["coef"][b] != 0).sum()))
    d.close()
