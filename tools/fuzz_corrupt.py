#!/usr/bin/env python3
"""Damaged codestreams: bit flips, cuts and garbage inside the entropy-coded data.  The decoder must
finish and report; status and samples are compared with the oracle's (the documented deviation -- the
block that is half read when the data runs out -- may differ, so sample mismatches are counted, not fatal).

    python tools/fuzz_corrupt.py [cases] [seed]
"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from jpeg_amd import binding
from jpeg_amd.synth import synth_image
from oracle.pyoracle import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
orc = Oracle()
st_diff = px_diff = 0
t0 = time.time()
for case in range(N):
    nc = rnd.choice([1, 3, 3])
    H, V = (rnd.choice([1, 2]), rnd.choice([1, 2])) if nc == 3 else (1, 1)
    X, Y = rnd.randint(16, 700), rnd.randint(16, 500)
    ri = rnd.choice([0, 0, 3, rnd.randint(1, 60)])
    img = synth_image(X, Y, nc, 8, 5000 * seed + case, False)
    if nc == 3 and rnd.randint(0, 3) == 0:
        # several scans (one per component, or groups): the damage may also hit the headers between them
        err, jpg = orc.encode_scans(img, P=8, H=H, V=V, q=rnd.randint(30, 97), optimize=rnd.randint(0, 1),
                                    restart_interval=ri, redefine=rnd.randint(0, 1), split=rnd.randint(0, 2))
    else:
        err, jpg, _ = orc.encode(img, P=8, H=H, V=V, q=rnd.randint(30, 97), optimize=rnd.randint(0, 1), restart_interval=ri)
    if err:
        continue
    sos = jpg.index(b"\xff\xda")
    start = sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])
    b = bytearray(jpg)
    kind = rnd.choice(["flip", "flip", "cut", "garbage", "ff"])
    if kind == "flip":
        for _ in range(rnd.randint(1, 8)):
            p = rnd.randint(start, len(b) - 3)
            b[p] ^= 1 << rnd.randint(0, 7)
    elif kind == "cut":
        p = rnd.randint(start, len(b) - 3)
        b = b[:p] + b[-2:] if rnd.randint(0, 1) else b[:p]
    elif kind == "garbage":
        p = rnd.randint(start, len(b) - 3)
        n = rnd.randint(1, 200)
        b[p:p + n] = bytes(rnd.randint(0, 255) for _ in range(n))
    else:
        p = rnd.randint(start, len(b) - 3)
        b[p:p + 2] = bytes([0xFF, rnd.choice([0x00, 0xD3, 0xFF, 0xE1, 0xD9, 0xC4])])
    bad = bytes(b)
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
        print("STATUS", case, kind, hex(st & 0xffffffff), hex(oerr & 0xffffffff), dict(X=X, Y=Y, nc=nc, H=H, V=V, ri=ri))
    elif st == 0 and not np.array_equal(px, opx):
        px_diff += 1
        frac = float((px != opx).any(axis=-1).mean()) if px.ndim == 3 else float((px != opx).mean())
        print("SAMPLES", case, kind, "differing pixels %.4f" % frac, dict(X=X, Y=Y, nc=nc, H=H, V=V, ri=ri), why)
print("corrupt fuzz: %d cases, %d status differences, %d sample differences, %.0f s" % (N, st_diff, px_diff, time.time() - t0))
