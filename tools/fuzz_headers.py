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
taken = st_diff = px_diff = big = n_wide = 0


def dc_categories_past_15(jpg):
    """does a DC table of the file, as parse_huffman_tables() reads it (decoder.c:143-184: the symbols that the
    sixteen counts announce, whatever the segment's length field says), hold a symbol of 16 or more?  Such a symbol is a
    category: the reference then reads that many extra bits and makes a difference beyond 16 bits of it
    (coeffs.c:33-48) -- deviation D4 whether or not the SUM comes back into range"""
    # (every FF C4 in front of the first scan: the reference finds its markers by looking for FF xx behind what its
    # parsers consumed, length fields or not -- io.c:196-210 -- and a stray FF C4 inside another segment is rare
    # enough for a classifier)
    sos = jpg.find(b"\xff\xda")
    i = jpg.find(b"\xff\xc4", 2)
    while 0 <= i < sos:
        ln = (jpg[i + 2] << 8) | jpg[i + 3]
        p = i + 4
        while p + 17 <= len(jpg) and p < i + 2 + ln:
            tc, counts = jpg[p] >> 4, jpg[p + 1:p + 17]
            n = sum(counts)
            if tc == 0 and any(v >= 16 for v in jpg[p + 17:p + 17 + n]):
                return True
            p += 17 + n
        i = jpg.find(b"\xff\xc4", i + 2)
    return False

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
    elif st == 0 and max(int(np.abs(c["coef"]).max()) for c in oinfo.comps) > 32767:
        n_wide += 1                                # D4: a damaged table made categories past 15 -- int32 coefficients in
                                                   # the reference, int16 here
    elif st == 0 and not np.array_equal(px, opx) and dc_categories_past_15(bad):
        n_wide += 1                                # D4 again: the difference left 16 bits, the sum came back
    elif st == 0 and not np.array_equal(px, opx):
        px_diff += 1
        frac = float((px != opx).mean())
        print("SAMPLES", case, f, "differing %.4f" % frac)
print("header fuzz: %d cases, %d taken by the parser (%d oversized skipped), %d status differences, %d sample differences, "
      "%d with coefficients beyond 16 bits (not compared), %.0f s" % (N, taken, big, st_diff, px_diff, n_wide, time.time() - t0))
