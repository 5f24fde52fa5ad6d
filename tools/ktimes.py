#!/usr/bin/env python3
"""Developer tool: per-kernel HIP-event times of one decode and one encode batch (no correctness
guard -- experiments may break results on purpose).  Usage: python tools/ktimes.py [config] [frames]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from jpeg_amd import binding, synth
    import bench
    name = sys.argv[1] if len(sys.argv) > 1 else "metric_4k_420_q90"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    torch.cuda.set_device(0)
    enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, name, F, 0, int(os.environ.get("KT_RI", "0")))
    dec = binding.Decoder(0)
    dec.prepare(jpgs)
    dec.set_debug(int(os.environ.get("KT_DEBUG", "0")))
    out = dec.alloc_output()
    for _ in range(2):
        dec.run(out)
        enc.run(samples)
    torch.cuda.synchronize()
    kd = bench.kernel_times(dec, lambda: dec.run(out), torch, reps)
    ke = bench.kernel_times(enc, lambda: enc.run(samples), torch, reps)
    pix = cfg["X"] * cfg["Y"] * F
    td = bench.timed(torch, lambda: dec.run(out), 10, 2, None) / 10
    te = bench.timed(torch, lambda: enc.run(samples), 10, 2, None) / 10
    dec.sync()
    stats = [dec.stats(f) for f in range(F)]
    try:
        print("chain steps, frame 0:", dec.chain_steps(0))
    except AttributeError:
        pass
    print("stats (serial, repaired, blocks, segments):", stats[:4], "serial frames", sum(s[0] for s in stats),
          "repaired wgs", sum(s[1] for s in stats))
    print(json.dumps({"config": name, "frames": F, "jpg_bytes": len(jpgs[0]),
                      "decode_ms": round(td * 1e3, 4), "encode_ms": round(te * 1e3, 4),
                      "decode_gpix_s": round(pix / td / 1e9, 2), "encode_gpix_s": round(pix / te / 1e9, 2),
                      "dec": {k: round(v, 4) for k, v in sorted(kd.items(), key=lambda kv: -kv[1])},
                      "enc": {k: round(v, 4) for k, v in sorted(ke.items(), key=lambda kv: -kv[1])}}))


if __name__ == "__main__":
    main()
