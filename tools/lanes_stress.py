#!/usr/bin/env python3
"""Developer tool: the decoder's lanes under repetition -- run() back to back without sync(), two output buffers
taking turns, a consumer kernel on the caller's stream behind every run(), batches of changing size and
prepare() in between: every output must check out against the one-lane result.  A race between lanes, or between a
lane that runs ahead and the caller's stream, shows up as a wrong checksum sooner or later.
Usage: python tools/lanes_stress.py [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from jpeg_amd import binding, synth
    import bench
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.cuda.set_device(0)
    bad = 0
    for name, F in (("c5_1080p_422_q85", 48), ("metric_4k_420_q90", 12), ("c2_4k_444_q90", 6)):
        enc, samples, jpgs, cfg = bench.make_batch(torch, binding, synth, name, F, 0, 480 if name.startswith("c2") else 0)
        enc.close()
        del samples
        ref = binding.Decoder(0)
        ref.prepare(jpgs)
        want = bench.frame_sums(torch, ref.run()).clone()
        assert all(s == 0 for s in ref.sync())
        ref.close()
        d = binding.Decoder(0)
        side = torch.cuda.Stream(0)
        with torch.cuda.stream(side):
            d.set_lanes(4)
            d.prepare(jpgs)
            outs = [d.alloc_output(), d.alloc_output()]
            got = []
            for r in range(rounds):
                n = F if r % 7 else max(1, F - 1 - r % 5)        # now and then a smaller batch, prepared anew
                if n != d.n:
                    if r:
                        d.sync()
                    d.prepare(jpgs[:n])
                o = outs[r & 1]
                o.fill_(r & 255)                                 # queued before run(): no lane may write ahead of it
                d.run(o)
                got.append((n, bench.frame_sums(torch, o[:n]).clone()))   # queued behind run() on the caller's stream
            d.sync()
        side.synchronize()
        for r, (n, g) in enumerate(got):
            if not torch.equal(g, want[:n]):
                bad += 1
                print("MISMATCH", name, "round", r, "frames", n, flush=True)
        d.close()
        print(name, "rounds", rounds, "bad so far", bad, flush=True)
    print("lanes stress: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
