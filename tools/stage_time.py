#!/usr/bin/env python3
"""Runs ON THE GPU BOX: per-stage time (HIP events, one batch alone, median of 15) of one resident batch.
usage: [CM3D_LIB=...] tools/stage_time.py config frames [config frames ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn

args = sys.argv[1:]
for name, F in zip(args[0::2], args[1::2]):
    F = int(F)
    import pickle
    cache = f"/tmp/cm3d_hb_{name}_{F}.pkl"          # the same batch for every variant of a sweep (generation takes longer than the passes)
    if os.path.exists(cache):
        hb = pickle.load(open(cache, "rb"))
    else:
        cfg = syn.config(name)
        frames = [syn.make_frame(cfg, i) for i in range(F)]
        lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
        hb = lifting.pack_frames(frames, lanes, [0] * F)
        del frames
        pickle.dump(hb, open(cache, "wb"), protocol=4)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    for _ in range(3):
        eng.run(masks="rle")
    torch.cuda.synchronize()
    eng.check_status()
    st = torch.cuda.current_stream().cuda_stream
    stages = [s for s in eng.STAGES if s != "sweeps"]
    calls = {"masks": lambda s: eng.stage_masks(s, "rle"), "project": eng.stage_sweep_project, "compact": eng.stage_compact,
             "medoid": eng.stage_medoid, "lanes": lambda s: (eng.wait_lane_grid(), eng.stage_lanes(s)), "boxes": eng.stage_boxes}
    ev = {s: [] for s in stages}
    for _ in range(15):
        eng.stage_begin(st)
        for s in stages:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); calls[s](st); b.record()
            ev[s].append((a, b))
    torch.cuda.synchronize()
    ms = {s: float(np.median([a.elapsed_time(b) for a, b in ev[s]])) for s in stages}
    print(f"{os.path.basename(os.environ.get('CM3D_LIB', 'product'))} {name} x{F}: total {sum(ms.values()) * 1e3:.0f} us  " +
          "  ".join(f"{k} {v * 1e3:.1f}" for k, v in ms.items()), flush=True)
