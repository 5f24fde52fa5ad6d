#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Run in the build container (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden.py

The reference ships no tests or vectors, so every golden here is
an output of the reference itself (oracle/_ref/libref_{enc,dec}.so = the unmodified
sources + tap harness).  Fixtures are DATA only: synthetic inputs are regenerated at
test time by jpeg_amd.synth (deterministic, integer-only), and the files hold
  * <name>.jpg      -- the reference encoder's output (small cases), or an externally
                       written JPEG (Pillow) used as decoder input,
  * manifest.json   -- per case: generator parameters, md5 of the reference JPEG, md5 of the
                       reference-decoded PPM/PGM body, md5 of the quantised coefficients.
"""
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jpeg_amd.synth import synth_image  # noqa: E402
from oracle.pyoracle import Oracle, Reference, pnm_body  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def md5(b):
    return hashlib.md5(b).hexdigest()


def coef_md5(img):
    h = hashlib.md5()
    for c in img.comps:
        h.update(np.ascontiguousarray(c["coef"], np.int32).tobytes())
    return h.hexdigest()


# name: X, Y, ncomp, P, H, V, q, optimize, range_safe, seed, keep_jpg
ENC_CASES = [
    ("e_444_q90_o0", 64, 48, 3, 8, 1, 1, 90, 0, False, 11, True),
    ("e_422_q85_o0", 96, 40, 3, 8, 2, 1, 85, 0, False, 12, True),
    ("e_420_q75_o1", 80, 64, 3, 8, 2, 2, 75, 1, False, 13, True),
    ("e_440_q50_o1", 48, 64, 3, 8, 1, 2, 50, 1, False, 14, True),
    ("e_420_unaligned_q75", 100, 75, 3, 8, 2, 2, 75, 1, False, 15, True),   # padding quirk Q7
    ("e_422_unaligned_q30", 37, 29, 3, 8, 2, 1, 30, 0, False, 16, True),
    ("e_gray8_q95_o1", 72, 56, 1, 8, 1, 1, 95, 1, False, 17, True),
    ("e_gray8_unaligned_q10", 33, 17, 1, 8, 1, 1, 10, 0, False, 18, True),
    ("e_gray12_q75_o0", 128, 64, 1, 12, 1, 1, 75, 0, True, 3, True),        # 12-bit, default tables
    ("e_gray12_q80_o1", 64, 64, 1, 12, 1, 1, 80, 1, True, 19, True),
    ("e_rgb12_420_q80_o1", 64, 48, 3, 12, 2, 2, 80, 1, True, 20, True),
    ("e_444_q100_o1", 40, 40, 3, 8, 1, 1, 100, 1, False, 21, True),
    ("e_420_q1_o1", 64, 64, 3, 8, 2, 2, 1, 1, False, 22, True),
    ("e_tiny_1x1", 1, 1, 3, 8, 2, 2, 75, 1, False, 23, True),
    ("e_tiny_gray_8x8", 8, 8, 1, 8, 1, 1, 75, 0, False, 24, True),
    # BASELINE config shapes: md5 only
    ("c1_512_420_q75_o0", 512, 512, 3, 8, 2, 2, 75, 0, False, 1234, False),
    ("c1_512_420_q75_o1", 512, 512, 3, 8, 2, 2, 75, 1, False, 1234, False),
    ("c2_480x270_444_q90", 480, 272, 3, 8, 1, 1, 90, 1, False, 2, False),
    ("metric_960x544_420_q90", 960, 544, 3, 8, 2, 2, 90, 0, False, 90, False),
    ("c3_960x544_gray12_q95", 960, 544, 1, 12, 1, 1, 95, 0, True, 3, False),
    ("c4_960x544_420_q95_o1", 960, 544, 3, 8, 2, 2, 95, 1, False, 4, False),
    ("c5_1080p_422_q85", 1920, 1080, 3, 8, 2, 1, 85, 0, False, 5, False),
]


def strip_dht(jpg):
    """Remove every DHT segment (Motion-JPEG style frame relying on implicit tables)."""
    out = bytearray(jpg[:2])
    pos = 2
    while pos < len(jpg):
        assert jpg[pos] == 0xFF
        m = jpg[pos + 1]
        ln = (jpg[pos + 2] << 8) | jpg[pos + 3]
        if m == 0xDA:
            out += jpg[pos:]
            break
        if m != 0xC4:
            out += jpg[pos:pos + 2 + ln]
        pos += 2 + ln
    return bytes(out)


def main():
    ref = Reference()
    orc = Oracle()
    manifest = {"encode": {}, "decode": {}}

    for (name, X, Y, nc, P, H, V, q, opt, safe, seed, keep) in ENC_CASES:
        img = synth_image(X, Y, nc, P, seed, safe)
        # the reference assert()s on |DC diff| > 2047 or a symbol missing from the default
        # tables, which would take this process down: pre-check with our port
        perr = orc.encode(img, P=P, H=H, V=V, q=q, optimize=opt)[0]
        assert perr == 0, (name, "input unsuitable for the reference encoder: 0x%x" % (perr & 0xffff))
        err, jpg, info = ref.encode(img, P=P, H=H, V=V, q=q, optimize=opt)
        assert err == 0, (name, err)
        err, dec, dinfo = ref.decode(jpg)
        assert err == 0, (name, err)
        entry = dict(X=X, Y=Y, ncomp=nc, P=P, H=H, V=V, q=q, optimize=opt, range_safe=safe, seed=seed,
                     input_md5=md5(pnm_body(img, P)), jpg_md5=md5(jpg), jpg_len=len(jpg),
                     coef_md5=coef_md5(info), dec_md5=md5(pnm_body(dec, P)))
        if keep:
            with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
                f.write(jpg)
            entry["jpg_file"] = name + ".jpg"
        manifest["encode"][name] = entry
        print(name, len(jpg))

    # ---- decoder-only inputs -------------------------------------------------------
    dec_inputs = {}
    # Motion-JPEG style: default tables, DHT stripped (common.c:90-99 implicit tables)
    img = synth_image(96, 64, 3, 8, 31, False)
    err, jpg, _ = ref.encode(img, P=8, H=2, V=1, q=85, optimize=0)
    dec_inputs["d_mjpeg_nodht_422"] = strip_dht(jpg)
    # trailing garbage after EOI and a second frame appended: only frame 0 is decoded
    dec_inputs["d_two_frames"] = jpg + jpg[:200]

    # one scan per component (non-interleaved order, decoder.c:274-302): written by the oracle's extension
    # writer, judged by the reference decoder like every other input here
    # and scans of SOME components (decoder.c:303-358 with Ns < Nf): split 1 = {Y} {Cb Cr}, 2 = {Y Cb} {Cr}
    for name, (X, Y, H, V, q, opt, ri, redef, seed, *split) in {
        "d_scans_420_opt": (80, 64, 2, 2, 80, 1, 0, 0, 51),
        "d_scans_444_rst7": (64, 48, 1, 1, 90, 0, 7, 0, 52),
        "d_scans_422_redefine": (96, 40, 2, 1, 70, 1, 0, 1, 53),
        "d_scans_420_unaligned_rst8": (100, 75, 2, 2, 80, 1, 8, 1, 54),
        # an interval that ends inside the reference's H x V block group: bug-compatible, not standard
        "d_scans_420_rst5_quirk": (100, 75, 2, 2, 80, 1, 5, 0, 55),
        "d_scans_420_y_cbcr": (80, 64, 2, 2, 80, 1, 0, 0, 56, 1),
        "d_scans_422_ycb_cr_rst3": (100, 75, 2, 1, 85, 0, 3, 0, 57, 2),
        "d_scans_420_ycb_cr_rst4_opt": (100, 75, 2, 2, 75, 1, 4, 0, 58, 2),
    }.items():
        err, data = orc.encode_scans(synth_image(X, Y, 3, 8, seed, False), P=8, H=H, V=V, q=q, optimize=opt,
                                     restart_interval=ri, redefine=redef, split=split[0] if split else 0)
        assert err == 0, (name, err)
        dec_inputs[name] = data

    try:
        from PIL import Image
        rgb = synth_image(112, 80, 3, 8, 41, False)
        im = Image.fromarray(rgb, "RGB")

        def pil(**kw):
            b = io.BytesIO()
            kw.setdefault("quality", 90)
            im.save(b, "JPEG", **kw)
            return b.getvalue()
        # restart markers every MCU row / every 2 MCUs, with libjpeg's own (optimised) tables
        dec_inputs["d_pil_444_rst_row"] = pil(subsampling=0, restart_marker_rows=1)
        dec_inputs["d_pil_420_rst_2mcu"] = pil(subsampling=2, restart_marker_blocks=2, optimize=True)
        dec_inputs["d_pil_422_jfif"] = pil(subsampling=1, quality=60)
        dec_inputs["d_pil_gray"] = (lambda b: (Image.fromarray(rgb[:, :, 1], "L").save(b, "JPEG", quality=80), b.getvalue())[1])(io.BytesIO())
        dec_inputs["d_pil_progressive"] = pil(progressive=True)          # must be rejected (SOF2)
        cmyk = Image.fromarray(np.dstack([rgb, rgb[:, :, 0]]), "CMYK")
        b = io.BytesIO()
        cmyk.save(b, "JPEG", quality=85)
        dec_inputs["d_pil_cmyk"] = b.getvalue()                          # 4 components (Adobe APP14)
    except ImportError:
        print("Pillow missing: external-encoder decode fixtures skipped")

    for name, data in dec_inputs.items():
        err, dec, dinfo = ref.decode(data)
        entry = dict(jpg_file=name + ".jpg", jpg_md5=md5(data), err=err)
        if err == 0:
            entry.update(dec_md5=md5(pnm_body(dec, dinfo.P)), coef_md5=coef_md5(dinfo),
                         X=dinfo.X, Y=dinfo.Y, Nf=dinfo.Nf, P=dinfo.P, shape=list(dec.shape))
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        manifest["decode"][name] = entry
        print(name, len(data), "err=0x%x" % err)

    # ---- block-level known-answer vectors from the reference's own primitives ----------
    blk = np.array([((i * 37 + 11) % 256) - 128 for i in range(64)], np.float32)
    f = ref.fdct(blk)
    manifest["kat"] = dict(
        lut_hex=["%08x" % v for v in ref.lut().view(np.uint32).ravel()],
        fdct_in="c[i]=((i*37+11)%256)-128",
        fdct_hex=["%08x" % v for v in f.view(np.uint32)],
        idct_hex=["%08x" % v for v in ref.idct(f).view(np.uint32)],
    )
    with open(os.path.join(HERE, "manifest.json"), "w") as fo:
        json.dump(manifest, fo, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
