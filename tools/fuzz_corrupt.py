#!/usr/bin/env python3
"""Damaged codestreams: bit flips, cuts and garbage inside the entropy-coded data.  The decoder must
finish and report; status and samples are compared with the oracle's.  The gating version of this run, with
the named deviations, is tests/test_gpu_damaged.py; this tool takes any count and seed.

    python tools/fuzz_corrupt.py [cases] [seed]
"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from jpeg_amd import binding
from oracle.pyoracle import Oracle
from tools.fuzz_cases import corrupt_case

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
orc = Oracle()
st_diff = px_diff = n_big = 0
t0 = time.time()
for case in range(N):
    bad, meta = corrupt_case(rnd, orc, seed, case)
    if bad is None:
        continue
    kind = meta["kind"]
    oerr, opx, oinfo = orc.decode(bad)
    d = binding.Decoder()
    try:
        d.prepare([bad])
        out = d.run()
        st = d.sync()[0]
        px = d.to_arrays(out)[0] if st == 0 else None
        why = ""
        if st == 0 and oerr == 0:
            for k, (a, c) in enumerate(zip(d.coefficients(0), oinfo.comps)):
                df = np.nonzero((a != c["coef"]).any(axis=1))[0]
                if len(df):
                    big = int(np.abs(c["coef"]).max())
                    why += " comp%d: %d/%d blocks differ, first %d, oracle |max| %d, mblocks %d;" % (k, len(df), len(a), df[0], big, oinfo.mblocks)
    except binding.JpegError as e:
        st, px = e.code, None
    d.close()
    if (st == 0) != (oerr == 0):
        st_diff += 1
        print("STATUS", case, kind, hex(st & 0xffffffff), hex(oerr & 0xffffffff), meta)
    elif st == 0 and max(int(np.abs(c["coef"]).max()) for c in oinfo.comps) > 32767:
        n_big += 1                                        # D4: int32 coefficients in the reference, int16 here
    elif st == 0 and not np.array_equal(px, opx):
        px_diff += 1
        frac = float((px != opx).any(axis=-1).mean()) if px.ndim == 3 else float((px != opx).mean())
        print("SAMPLES", case, kind, "differing pixels %.4f" % frac, meta, why)
print("corrupt fuzz: %d cases, %d status differences, %d sample differences, %d with coefficients beyond 16 bits (not compared), %.0f s" % (
    N, st_diff, px_diff, n_big, time.time() - t0))
