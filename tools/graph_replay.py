#!/usr/bin/env python3
"""Eager passes against HIP-graph replays (LiftEngine.capture_graph) for several batch sizes: a small batch is
launch-bound, and one graph launch replaces ~13 kernel launches plus the Python between them."""
import sys
import time
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config("c2")
for F in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 256]:
    frames = [syn.make_frame(cfg, i) for i in range(F)]
    lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    for _ in range(3):
        eng.run(masks="rle")
    g = eng.capture_graph(masks="rle")
    n = 300
    out = []
    for fn in (lambda: eng.run(masks="rle"), g.replay):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n)
    print(f"{F:4d} frames: eager {out[0] * 1e6:7.1f} us/pass ({F / out[0]:9.0f} frames/s)   graph {out[1] * 1e6:7.1f} us/pass ({F / out[1]:9.0f} frames/s)")
