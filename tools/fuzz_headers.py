#!/usr/bin/env python3
"""Damaged HEADERS: random byte changes between SOI and the end of the first scan header of the golden streams.
Where the host parser still takes the stream, the GPU decode must finish and agree with the oracle on status
and samples (tables, sampling factors, table selectors and sizes are whatever the damage made of them).

    python tools/fuzz_headers.py [cases] [seed]
"""
import os, sys, random, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from jpeg_amd import binding
from oracle.pyoracle import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
orc = Oracle()
G = os.path.join(ROOT, "tests", "golden")
man = json.load(open(os.path.join(G, "manifest.json")))
files = [g["jpg_file"] for k in ("decode", "encode") for g in man[k].values() if "jpg_file" in g and g.get("err", 0) == 0]
data = {f: open(os.path.join(G, f), "rb").read() for f in files}
taken = st_diff = px_diff = big = 0
t0 = time.time()
for case in range(N):
    f = rnd.choice(files)
    jpg = bytearray(data[f])
    sos = jpg.index(b"\xff\xda")
    for _ in range(rnd.randint(1, 3)):
        p = rnd.randint(2, sos + 13)
        jpg[p] = rnd.randint(0, 255) if rnd.randint(0, 1) else jpg[p] ^ (1 << rnd.randint(0, 7))
    bad = bytes(jpg)
    err, fi = binding.parse_info(bad)
    if err:
        continue
    if fi.X * fi.Y > 4000 * 4000:
        big += 1                                   # a damaged size field: not this tool's subject
        continue
    taken += 1
    print("case", case, f, fi.X, fi.Y, fi.Nf, flush=True)
    oerr, opx, oinfo = orc.decode(bad)
    d = binding.Decoder()
    try:
        d.prepare([bad])
        out = d.run()
        st = d.sync()[0]
        px = d.to_arrays(out)[0] if st == 0 else None
    except binding.JpegError as e:
        st, px = e.code, None
    d.close()
    if (st == 0) != (oerr == 0):
        st_diff += 1
        print("STATUS", case, f, hex(st & 0xffffffff), hex(oerr & 0xffffffff))
    elif st == 0 and not np.array_equal(px, opx):
        px_diff += 1
        frac = float((px != opx).mean())
        print("SAMPLES", case, f, "differing %.4f" % frac)
print("header fuzz: %d cases, %d taken by the parser (%d oversized skipped), %d status differences, %d sample differences, %.0f s"
      % (N, taken, big, st_diff, px_diff, time.time() - t0))
