#!/usr/bin/env python3
"""Wall time of LiftEngine.run() over the default batch (or `config frames`), for A/B runs under environment knobs
(CM3D_FUSED_SWEEPS=0/1, ...).  Prints ms per pass."""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import lifting, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = syn.config(name)
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
for _ in range(10):
    eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    eng.run(masks="rle")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"{name} x{F}: fused_sweeps={eng.can_fuse_sweeps()} {dt * 1e3:.4f} ms/pass, {F / dt:.0f} frames/s")
