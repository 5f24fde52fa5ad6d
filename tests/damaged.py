"""Hand-made damaged codestreams for the parity tests (shared by the CPU and the GPU tests; data only)."""
import numpy as np


class BitWriter:
    """entropy-coded segment: MSB first, FF -> FF 00, padded with 1s (io.c:43-87, 277-290)"""

    def __init__(self):
        self.bits = []

    def put(self, code, size):
        self.bits += [(code >> (size - 1 - i)) & 1 for i in range(size)]

    def bytes(self):
        b = self.bits + [1] * (-len(self.bits) % 8)
        out = bytearray()
        for i in range(0, len(b), 8):
            v = int("".join(map(str, b[i:i + 8])), 2)
            out.append(v)
            if v == 0xFF:
                out.append(0)
        return bytes(out)


def extra_bits(v, cat):
    return (v if v >= 0 else v - 1) & ((1 << cat) - 1)


def zero_run_past_block_end(oracle, value_cat=0):
    """A gray 24x8 frame (three blocks, default tables) whose first block holds a zero run that ends past
    coefficient 63 (coeffs.c:246-251): DC 0, ZRL x 3 (coefficient pointer at 49), then a symbol with run 15
    -- ZRL again (value_cat = 0: the stray store writes a zero) or (15, value_cat) with value +1.
    The block has no EOB: it is complete because `rem` dropped below 1.  Blocks 1 and 2 are ordinary.
    -> (codestream, expected coefficients of blocks 1 and 2 as {raster index: value})"""
    img = np.full((8, 24), 128, np.uint8)
    err, jpg, _ = oracle.encode(img, P=8, H=1, V=1, q=50, optimize=0)
    assert err == 0
    sos = jpg.index(b"\xff\xda")
    head = jpg[:sos + 2 + ((jpg[sos + 2] << 8) | jpg[sos + 3])]
    dcs, dcc = oracle.make_codes(*oracle.default_table(0, 0))
    acs, acc = oracle.make_codes(*oracle.default_table(1, 0))
    w = BitWriter()
    dc = lambda diff: (w.put(int(dcc[abs(diff).bit_length()]), int(dcs[abs(diff).bit_length()])),
                       w.put(extra_bits(diff, abs(diff).bit_length()), abs(diff).bit_length()) if diff else None)
    ac = lambda run, v: (w.put(int(acc[(run << 4) | abs(v).bit_length()]), int(acs[(run << 4) | abs(v).bit_length()])),
                         w.put(extra_bits(v, abs(v).bit_length()), abs(v).bit_length()) if v else None)
    dc(0)
    for _ in range(3):
        ac(15, 0)
    if value_cat:
        ac(15, 1)                      # 49 + 15 = 64: past the end, value +1
    else:
        ac(15, 0)
    dc(5)                              # block 1: DC 5, first AC 3, EOB
    ac(0, 3)
    ac(0, 0)
    dc(-2)                             # block 2: DC 5 - 2 = 3, EOB
    ac(0, 0)
    return head + w.bytes() + b"\xff\xd9", {1: {0: 5, 1: 3}, 2: {0: 3}}


def deep_image(P, nc, X=160, Y=96, seed=1):
    """A low-contrast image around mid grey with P bits per sample (13 ... 16): what the reference encoder takes at
    such precisions, where its assert on |DC| <= 2047 (encoder.c:447, :501) also holds for the first block."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:Y, 0:X]
    a = (1 << (P - 1)) + ((np.sin(xx / 9.0) + np.cos(yy / 7.0)) * (1 << (P - 6))).astype(np.int64)
    a = a + rng.integers(-(1 << (P - 9)), 1 << (P - 9), size=(Y, X))
    a = np.clip(a, 0, (1 << P) - 1).astype(np.uint16)
    if nc == 3:
        a = np.stack([a, np.roll(a, 5, 1), np.roll(a, 9, 0)], -1)
    return a


def table_per_component(jpg):
    """A single-scan file rewritten so that the j-th component of its scan reads DC table j and AC table j (copies of
    the tables it named): with four components all eight table ids of decoder.c:176-182 are in use at once, the most
    look-up tables a scan can ask the kernels to hold.  The samples it decodes to stay what they were."""
    segs, i = [], 2
    while jpg[i + 1] != 0xDA:
        n = (jpg[i + 2] << 8) | jpg[i + 3]
        segs.append((jpg[i + 1], jpg[i + 4:i + 2 + n]))
        i += 2 + n
    tables = {}
    for m, body in segs:
        p = 0
        while m == 0xC4 and p < len(body):
            n = sum(body[p + 1:p + 17])
            tables[body[p] >> 4, body[p] & 15] = body[p + 1:p + 17 + n]
            p += 17 + n
    n = (jpg[i + 2] << 8) | jpg[i + 3]
    sos = bytearray(jpg[i + 4:i + 2 + n])
    dht = b""
    for j in range(sos[0]):
        td, ta = sos[2 + 2 * j] >> 4, sos[2 + 2 * j] & 15
        dht += bytes([j]) + tables[0, td] + bytes([0x10 | j]) + tables[1, ta]
        sos[2 + 2 * j] = (j << 4) | j
    out = jpg[:2]
    for m, body in segs:
        if m != 0xC4:
            out += bytes([0xFF, m]) + (len(body) + 2).to_bytes(2, "big") + body
    out += b"\xff\xc4" + (len(dht) + 2).to_bytes(2, "big") + dht
    out += b"\xff\xda" + n.to_bytes(2, "big") + bytes(sos)
    return out + jpg[i + 2 + n:]
