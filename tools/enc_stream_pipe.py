#!/usr/bin/env python3
"""Developer tool: frames/s of mjpeg.encode_stream on the 1080p stream (single rank)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import binding, synth, mjpeg
import bench
torch.cuda.set_device(0)
r = bench.mjpeg_encode_stream(torch, binding, synth, mjpeg, 0, 0, 1, None, 256)
print(r["passes_frames_per_s"])
