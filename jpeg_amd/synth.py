"""Deterministic, integer-only synthetic test images.

The survey's recipe is  v = 0.5 + 0.25*sin(..) + 0.15*cos(..) + noise ; it recommends an
integer-only generator so that inputs are bit-reproducible on any box.  This module keeps the
recipe's shape (two plane waves per channel plus small noise) but evaluates it entirely in
int64: phases in 1/65536 turns, a parabolic sine with 15 fractional bits, and a 32-bit
integer hash for the noise.  No floating point, no libm.
"""
import numpy as np

CONFIGS = {
    # name: (X, Y, ncomp, P, H, V, q, optimize, range_safe, seed)        SURVEY.md section 8 table
    "c1_512_420_q75":   (512, 512, 3, 8, 2, 2, 75, 0, False, 1234),
    "c2_4k_444_q90":    (3840, 2160, 3, 8, 1, 1, 90, 1, False, 2),
    "metric_4k_420_q90": (3840, 2160, 3, 8, 2, 2, 90, 0, False, 90),
    "c3_8k_gray12_q95": (7680, 4320, 1, 12, 1, 1, 95, 0, True, 3),
    "c4_8k_420_q95":    (7680, 4320, 3, 8, 2, 2, 95, 1, False, 4),
    "c5_1080p_422_q85": (1920, 1080, 3, 8, 2, 1, 85, 0, False, 5),
}


def _isin(ph):
    """Parabolic sine: ph in 1/65536 turns (int64 array) -> value in [-32768, 32768]."""
    ph = ph & 0xFFFF
    half = ph & 0x7FFF                      # position inside the half period, 0..32767
    mag = (half * (32768 - half)) >> 13     # 0 .. 32768 at the quarter point
    return np.where(ph & 0x8000, -mag, mag)


def _hash32(x):
    """32-bit integer mix (xorshift-multiply), vectorised on uint64 holding 32-bit values."""
    m = np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint64) & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x.astype(np.int64)


def synth_image(X, Y, ncomp=3, P=8, seed=0, range_safe=False, frame=0):
    """-> (Y, X) or (Y, X, 3) array, uint8 (P <= 8) or uint16.

    range_safe selects the low-contrast variant needed by the reference encoder's
    |DC diff| <= 2047 assert on 12-bit q95 material: smaller amplitudes and
    the SAME horizontal phase at x = 0 and x = X (periodic rows), mean at mid-grey.
    """
    maxval = (1 << P) - 1
    yy, xx = np.meshgrid(np.arange(Y, dtype=np.int64), np.arange(X, dtype=np.int64), indexing="ij")
    out = np.empty((Y, X, ncomp), dtype=np.int64)
    for k in range(ncomp):
        a = 3 * (k + 1) if not range_safe else 3
        b = 5 * (k + 2) if not range_safe else 5
        # frames differ by a phase shift and by their noise; the range-safe variant keeps the phase (the first
        # block's DC and every DC difference must stay inside +-2047) and differs by noise only
        ph1 = (a * xx * 65536) // X + (2 * yy * 65536) // Y + (0 if range_safe else 4096 * frame)
        ph2 = (7 * xx * 65536) // X - (b * yy * 65536) // Y
        if not range_safe:
            ph2 = ph2 + 16384                                    # +1/4 turn: cosine
        s1, s2 = _isin(ph1), _isin(ph2)
        h = _hash32((xx * 0x9E3779B1 + yy * 0x85EBCA77 + (k + 1) * 0xC2B2AE3D
                     + (seed + 1) * 0x27D4EB2F + frame * 0x165667B1))
        # sum of four bytes - 510: triangular-ish noise in [-510, 510], sigma ~ 148
        noise = ((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF)) - 510
        if range_safe:
            # 0.5 + 0.2*s1 + 0.1*s2 + ~0.01 noise
            v = 32768 * 50 + 20 * s1 + 10 * s2 + (noise * 32768 * 1) // 148
        else:
            # 0.5 + 0.25*s1 + 0.15*s2 + ~0.03 noise        (scale: 100 * 32768 == 1.0)
            v = 32768 * 50 + 25 * s1 + 15 * s2 + (noise * 32768 * 3) // 148
        v = (v * maxval + 50 * 32768) // (100 * 32768)
        out[:, :, k] = np.clip(v, 0, maxval)
    out = out.astype(np.uint8 if P <= 8 else np.uint16)
    return out[:, :, 0] if ncomp == 1 else out


def config_image(name, scale=1, frame=0):
    """Image of a named BASELINE config; scale > 1 shrinks both dimensions (parity-test sizes)."""
    X, Y, nc, P, H, V, q, opt, safe, seed = CONFIGS[name]
    return synth_image(X // scale, Y // scale, nc, P, seed, safe, frame), dict(P=P, H=H, V=V, q=q, optimize=opt)


def synth_body_torch(X, Y, ncomp=3, P=8, seed=0, range_safe=False, frame=0, device="cuda"):
    """The same image as synth_image(), computed with torch int64 on `device`, as the PPM/PGM BODY the
    encoder reads: uint8 tensor of Y*X*ncomp bytes (P <= 8) or Y*X*ncomp big-endian 16-bit samples.
    Integer-only like the numpy version, so the two are bit-identical (tests/test_gpu_parity.py); bench.py
    uses it to make a batch of distinct frames without minutes of host arithmetic."""
    import torch
    maxval = (1 << P) - 1
    yy = torch.arange(Y, dtype=torch.int64, device=device).view(Y, 1)
    xx = torch.arange(X, dtype=torch.int64, device=device).view(1, X)

    def isin(ph):
        ph = ph & 0xFFFF
        half = ph & 0x7FFF
        mag = (half * (32768 - half)) >> 13
        return torch.where((ph & 0x8000) != 0, -mag, mag)

    def hash32(x):
        m = 0xFFFFFFFF
        x = x & m
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & m
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & m                               # wraps in int64: the low 32 bits are what counts
        x = x ^ (x >> 16)
        return x

    fdiv = lambda a, b: torch.div(a, b, rounding_mode="floor")
    planes = []
    for k in range(ncomp):
        a = 3 * (k + 1) if not range_safe else 3
        b = 5 * (k + 2) if not range_safe else 5
        ph1 = fdiv(a * xx * 65536, X) + fdiv(2 * yy * 65536, Y) + (0 if range_safe else 4096 * frame)
        ph2 = fdiv(7 * xx * 65536, X) - fdiv(b * yy * 65536, Y)
        if not range_safe:
            ph2 = ph2 + 16384
        s1, s2 = isin(ph1), isin(ph2)
        h = hash32(xx * 0x9E3779B1 + yy * 0x85EBCA77 + (k + 1) * 0xC2B2AE3D + (seed + 1) * 0x27D4EB2F + frame * 0x165667B1)
        noise = ((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF)) - 510
        if range_safe:
            v = 32768 * 50 + 20 * s1 + 10 * s2 + fdiv(noise * 32768 * 1, 148)
        else:
            v = 32768 * 50 + 25 * s1 + 15 * s2 + fdiv(noise * 32768 * 3, 148)
        v = fdiv(v * maxval + 50 * 32768, 100 * 32768)
        planes.append(torch.clamp(v, 0, maxval))
    img = torch.stack(planes, dim=2)                           # [Y, X, ncomp] int64
    if P <= 8:
        return img.to(torch.uint8).reshape(-1)
    return torch.stack([(img >> 8) & 0xFF, img & 0xFF], dim=3).to(torch.uint8).reshape(-1)
