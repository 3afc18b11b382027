#!/usr/bin/env python3
"""Times cm3d_lane_nn alone on the default batch for a few lane-table seeds and prints the distance distribution."""
import sys
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config("c2")
F = 256
frames = [syn.make_frame(cfg, i) for i in range(F)]
for seed in [int(a) for a in sys.argv[1:]] or [7, 1]:
    lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=seed, extent=260.0)]
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.stage_lanes(st)
    b.record()
    torch.cuda.synchronize()
    d = eng.b.lane_dist.cpu().numpy()
    ok = np.isfinite(d)
    print(f"seed {seed}: lane_nn {a.elapsed_time(b) / 10 * 1e3:.1f} us; centroids {int(ok.sum())}; distance percentiles 50/90/99/max "
          f"{np.percentile(d[ok], [50, 90, 99]).round(1)} {d[ok].max():.1f}; beyond 40 m: {int((d[ok] > 40).sum())}")
