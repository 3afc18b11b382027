#!/usr/bin/env python3
"""Passes over N independent resident batches issued round-robin on N streams (one LiftEngine each): how much of the
latency-bound tail of a pass (lane search, boxes, scans) hides under another batch's heavy kernels.
usage: pass_time2.py [config] [frames] [n_engines]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import lifting, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
E = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = syn.config(name)
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
engs, streams = [], []
for e in range(E):
    frames = [syn.make_frame(cfg, e * F + i) for i in range(F)]
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng = lifting.LiftEngine()
        eng.upload(hb)
        for _ in range(5):
            eng.run(masks="rle")
    engs.append(eng); streams.append(s)
torch.cuda.synchronize()
for eng in engs:
    eng.check_status()
K = 200
t0 = time.perf_counter()
for k in range(K):
    with torch.cuda.stream(streams[k % E]):
        engs[k % E].run(masks="rle")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"{name} x{F}, {E} batches in flight: {dt * 1e3:.4f} ms/pass, {F / dt:.0f} frames/s")
