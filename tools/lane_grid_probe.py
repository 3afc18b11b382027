#!/usr/bin/env python3
"""Times cm3d_lane_grid_build for a few synthetic lane tables and prints the grid geometry it chose."""
import struct
import sys
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config("tiny")
frames = [syn.make_frame(cfg, i) for i in range(2)]
for seed in [int(a) for a in sys.argv[1:]] or [1, 7, 8, 9, 10, 11, 12, 13, 14]:
    lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=seed, extent=260.0)]
    hb = lifting.pack_frames(frames, lanes, [0, 0])
    eng = lifting.LiftEngine()
    eng.upload(hb)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        eng.stage_lane_grid(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.stage_lane_grid(st)
    b.record()
    torch.cuda.synchronize()
    g = eng.b.grid[:32].cpu().numpy().tobytes()
    x0, y0, h, inv_h, gw, gh, cell_base, margin = struct.unpack("ffffiiif", g)
    yaw = lanes[0][:, 2]
    along_x = float(np.mean(np.abs(np.cos(yaw)) > 0.7))
    print(f"seed {seed}: build {a.elapsed_time(b) / 10 * 1e3:.1f} us  gw {gw} gh {gh} h {h}  share of points on x-aligned lanes {along_x:.2f}")
