#!/usr/bin/env python3
"""Developer tool: decode rate of one batch laid out over K decoder objects with a stream each, K = 1, 2, 3, 4
(the narrow kernels of one part beside the wide ones of the others).  Usage: python tools/multi_dec.py [config] [frames]"""
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
    enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, name, F, 0, int(os.environ.get("KT_RI", "0")))
    pix = cfg["X"] * cfg["Y"] * F
    res = {"config": name, "frames": F}
    for K in (1, 2, 3, 4, 1):
        per = (F + K - 1) // K
        parts = [jpgs[k:k + per] for k in range(0, F, per)]
        decs, outs = [], []
        for js in parts:
            d = binding.Decoder(0)
            d.prepare(js)
            decs.append(d)
            outs.append(d.alloc_output())
        streams = binding.StreamSet(0, len(parts)) if K > 1 else None
        ss = streams.streams if K > 1 else [torch.cuda.current_stream()]

        def run():
            for d, o, s in zip(decs, outs, ss):
                with torch.cuda.stream(s):
                    d.run(o)

        run()
        assert all(s == 0 for d in decs for s in d.sync())
        t = bench.timed(torch, run, 10, 2, None) / 10
        res.setdefault("gpix_s", []).append((K, round(pix / t / 1e9, 2)))
        for d in decs:
            d.close()
        del outs
        if streams is not None:
            streams.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
