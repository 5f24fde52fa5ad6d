#!/usr/bin/env python3
"""Experiment: the 64-frame step split over several decoder/encoder objects, each on a stream of its own."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth
from bench import make_batch, timed

F = 64
for parts in (1, 2, 4, 8):
    n = F // parts
    encs, samples, jpgs, imgs, cfg = [], [], None, None, None
    decs, outs, streams = [], [], []
    for p in range(parts):
        e, s, j, im, cfg = make_batch(binding, synth, "metric_4k_420_q90", n, 0)
        d = binding.Decoder(0)
        d.prepare(j)
        encs.append(e); samples.append(s); decs.append(d); outs.append(d.alloc_output())
        streams.append((torch.cuda.Stream(0), torch.cuda.Stream(0)))

    def step():
        for p in range(parts):
            with torch.cuda.stream(streams[p][0]):
                decs[p].run(outs[p])
            with torch.cuda.stream(streams[p][1]):
                encs[p].run(samples[p])

    def step_dec():
        for p in range(parts):
            with torch.cuda.stream(streams[p][0]):
                decs[p].run(outs[p])

    def step_enc():
        for p in range(parts):
            with torch.cuda.stream(streams[p][1]):
                encs[p].run(samples[p])
    pix = cfg["X"] * cfg["Y"]
    r = {"parts": parts}
    for name, fn in (("both", step), ("dec", step_dec), ("enc", step_enc)):
        t = timed(torch, fn, 10, 3, None)
        r[name + "_ms"] = round(t / 10 * 1e3, 3)
        r[name + "_gpix"] = round(F * pix * 10 / t / 1e9, 1)
    print(json.dumps(r))
    for d in decs: d.close()
    for e in encs: e.close()
    del outs, samples
