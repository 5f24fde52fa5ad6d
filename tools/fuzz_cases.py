"""The damaged-stream generator shared by tools/fuzz_corrupt.py, tools/dbg_corrupt.py and
tests/test_gpu_damaged.py: one case = one synthetic image, encoded by the oracle, damaged inside its
entropy-coded data (bit flips, a cut, garbage bytes, a stray FF xx)."""
from jpeg_amd.synth import synth_image


def corrupt_case(rnd, orc, seed, case):
    """-> (damaged codestream or None, description)"""
    nc = rnd.choice([1, 3, 3])
    H, V = (rnd.choice([1, 2]), rnd.choice([1, 2])) if nc == 3 else (1, 1)
    X, Y = rnd.randint(16, 700), rnd.randint(16, 500)
    ri = rnd.choice([0, 0, 3, rnd.randint(1, 60)])
    img = synth_image(X, Y, nc, 8, 5000 * seed + case, False)
    scans = nc == 3 and rnd.randint(0, 3) == 0
    if scans:
        # several scans (one per component, or groups): the damage may also hit the headers between them
        err, jpg = orc.encode_scans(img, P=8, H=H, V=V, q=rnd.randint(30, 97), optimize=rnd.randint(0, 1),
                                    restart_interval=ri, redefine=rnd.randint(0, 1), split=rnd.randint(0, 2))
    else:
        err, jpg, _ = orc.encode(img, P=8, H=H, V=V, q=rnd.randint(30, 97), optimize=rnd.randint(0, 1), restart_interval=ri)
    if err:
        return None, None
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
    return bytes(b), dict(kind=kind, X=X, Y=Y, nc=nc, H=H, V=V, ri=ri, scans=scans, start=start, at=p, length=len(b))
