#!/usr/bin/env python3
"""Developer tool: wall-clock time per invocation of the reference's own programs, with the reference's loops
(oracle/_ref/decoder, encoder) and on libjpeg_amd.so (decoder_hip / encoder_hip: read_ecs, epilogue, prologue,
write_ecs from integration/ref_glue.c; *_stages: the arithmetic stages one by one), on one 4K 4:2:0 q90 image.
One process per file, as the reference is used: the HIP runtime's start-up is part of every run.
Usage: python tools/glue_time.py [X Y]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from jpeg_amd.synth import synth_image
    from oracle.pyoracle import pnm_bytes
    X = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
    Y = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
    ref = os.path.join(ROOT, "oracle", "_ref")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "jpeg_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    img = synth_image(X, Y, 3, 8, seed=7)
    res = {"image": "%dx%d 8-bit RGB, 4:2:0, q90" % (X, Y)}
    with tempfile.TemporaryDirectory() as d:
        ppm = os.path.join(d, "in.ppm")
        with open(ppm, "wb") as f:
            f.write(pnm_bytes(img, 8))

        def timed(prog, args, out):
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([os.path.join(ref, prog)] + args, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                t = time.perf_counter() - t0
                assert r.returncode == 0, (prog, r.stdout[-400:])
                best = t if best is None else min(best, t)
            with open(os.path.join(d, out), "rb") as f:
                return best, f.read()

        enc_args = ["-h", "2", "-v", "2", "-q", "90", "in.ppm"]
        t_ref, jpg = timed("encoder", enc_args + ["ref.jpg"], "ref.jpg")
        res["encoder_s"] = {"reference": round(t_ref, 3)}
        for prog in ("encoder_hip", "encoder_stages"):
            t, j = timed(prog, enc_args + [prog + ".jpg"], prog + ".jpg")
            assert j == jpg, prog
            res["encoder_s"][prog] = round(t, 3)
        t_ref, out = timed("decoder", ["ref.jpg", "ref.ppm"], "ref.ppm")
        res["decoder_s"] = {"reference": round(t_ref, 3)}
        for prog in ("decoder_hip", "decoder_stages"):
            t, o = timed(prog, ["ref.jpg", prog + ".ppm"], prog + ".ppm")
            assert o == out, prog
            res["decoder_s"][prog] = round(t, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
