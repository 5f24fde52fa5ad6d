#!/usr/bin/env python3
"""Developer tool: encode rate of one batch laid out over K encoder objects with a stream each, K = 1, 2, 3, 4.
Usage: python tools/multi_enc.py [config] [frames]"""
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
    torch.cuda.set_device(0)
    ri = int(os.environ.get("KT_RI", "0"))
    enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, name, F, 0, ri)
    enc.close()
    pix = cfg["X"] * cfg["Y"] * F
    res = {"config": name, "frames": F}
    for K in (1, 2, 3, 4, 1):
        per = (F + K - 1) // K
        parts = [samples[k:k + per] for k in range(0, F, per)]
        encs = []
        for p in parts:
            e = binding.Encoder(0)
            e.configure(p.shape[0], cfg["X"], cfg["Y"], cfg["nc"], cfg["P"], cfg["H"], cfg["V"], cfg["q"], cfg["opt"], ri)
            encs.append(e)
        streams = binding.StreamSet(0, len(parts)) if K > 1 else None
        ss = streams.streams if K > 1 else [torch.cuda.current_stream()]

        def run():
            for e, p, s in zip(encs, parts, ss):
                with torch.cuda.stream(s):
                    e.run(p)

        run()
        assert all(s == 0 for e in encs for s in e.sync()[0])
        t = bench.timed(torch, run, 10, 2, None) / 10
        res.setdefault("gpix_s", []).append((K, round(pix / t / 1e9, 2)))
        for e in encs:
            e.close()
        if streams is not None:
            streams.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
