#!/usr/bin/env python3
"""Randomised parity run: HIP path (through the C-ABI) against the CPU oracle, bit-exact.

    python tools/fuzz.py [cases] [seed]

Every case draws a geometry (size, components, sampling, precision), a quality, table mode and restart
interval, encodes a small batch on the GPU and with the oracle (codestream bytes must be identical), and
decodes the oracle's stream -- interleaved and as one scan per component -- on the GPU (coefficients and
samples must be identical).  Test infrastructure: uses oracle/.
"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from jpeg_amd import binding
from jpeg_amd.synth import synth_image
from oracle.pyoracle import Oracle
from tests.damaged import deep_image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
BIG = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # size multiplier (streams of many workgroups)
rnd = random.Random(seed)
orc = Oracle()
t0 = time.time()
bad = 0
skipped = 0
for case in range(N):
    nc = rnd.choice([1, 3, 3, 3])
    P = rnd.choice([8, 8, 8, 12, 12, 10, 5, 14, 16])
    H, V = (rnd.choice([1, 2]), rnd.choice([1, 2])) if nc == 3 else (1, 1)
    X, Y = rnd.choice([rnd.randint(1, 40), rnd.randint(40, 400 * BIG)]), rnd.choice([rnd.randint(1, 40), rnd.randint(40, 300 * BIG)])
    q = rnd.choice([rnd.randint(1, 100), rnd.randint(60, 98)])
    opt = rnd.randint(0, 1)
    ri = rnd.choice([0, 0, 1, rnd.randint(2, 40)])
    nfr = rnd.choice([1, 1, 2, 5])
    desc = dict(case=case, X=X, Y=Y, nc=nc, P=P, H=H, V=V, q=q, opt=opt, ri=ri, nfr=nfr)
    if P > 12:
        # what the reference's DC assert leaves at such precisions: low contrast around mid grey (tests/damaged.py)
        imgs = [deep_image(P, nc, X, Y, seed=1000 * seed + 10 * case + i) for i in range(nfr)]
        q = min(q, rnd.randint(20, 60))
        desc["q"] = q
    else:
        imgs = [synth_image(X, Y, nc, P, 1000 * seed + 10 * case + i, P >= 10, frame=i) for i in range(nfr)]
    refs = [orc.encode(im, P=P, H=H, V=V, q=q, optimize=opt, restart_interval=ri) for im in imgs]
    # ---- encode
    e = binding.Encoder()
    e.configure(nfr, X, Y, nc, P, H, V, q, opt, restart_interval=ri)
    e.run(e.upload(imgs))
    st, ln = e.sync()
    for i in range(nfr):
        oerr, ojpg, oinfo = refs[i]
        if (st[i] & 0xffffffff) != (oerr & 0xffffffff):
            print("ENCODE STATUS", desc, i, hex(st[i] & 0xffffffff), hex(oerr & 0xffffffff)); bad += 1
        elif oerr == 0 and e.jpeg(i, ln[i]) != ojpg:
            print("ENCODE BYTES", desc, i); bad += 1
    e.close()
    good = [r[1] for r in refs if r[0] == 0]
    if not good:
        skipped += 1
        continue
    # ---- decode, interleaved
    d = binding.Decoder()
    d.set_debug(rnd.choice([0, 0, 1, 2]))
    d.prepare(good)
    out = d.run()
    stt = d.sync()
    arrs = d.to_arrays(out)
    for i, j in enumerate(good):
        oerr, opx, oinfo = orc.decode(j)
        if stt[i] != oerr or not np.array_equal(arrs[i], opx):
            print("DECODE", desc, i, stt[i], oerr); bad += 1
        else:
            for a, c in zip(d.coefficients(i), oinfo.comps):
                if not np.array_equal(a, c["coef"]):
                    print("DECODE COEFS", desc, i); bad += 1
                    break
    d.close()
    # ---- decode, one scan per component
    if nc == 3 and P == 8:
        blocks = H * V
        split = rnd.randint(0, 2)                       # one scan per component, {Y}{Cb Cr}, {Y Cb}{Cr}
        rib = rnd.choice([0, blocks * rnd.randint(1, 9), rnd.randint(1, 11)])
        err, sj = orc.encode_scans(imgs[0], P=P, H=H, V=V, q=q, optimize=opt, restart_interval=rib, redefine=rnd.randint(0, 1), split=split)
        if err == 0:
            oerr, opx, oinfo = orc.decode(sj)
            d = binding.Decoder()
            d.prepare([sj])
            out = d.run()
            stt = d.sync()
            if stt[0] != oerr or not np.array_equal(d.to_arrays(out)[0], opx):
                print("DECODE SCANS", desc, rib, split, stt[0], oerr); bad += 1
            d.close()
    if case % 20 == 19:
        print("... %d cases, %d mismatches, %d without a valid stream, %.0f s" % (case + 1, bad, skipped, time.time() - t0), flush=True)
print("fuzz: %d cases, %d mismatches, %d skipped" % (N, bad, skipped))
sys.exit(1 if bad else 0)
