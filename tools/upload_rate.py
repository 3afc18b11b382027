#!/usr/bin/env python3
"""PCIe-inclusive rate of the default batch: host buffers -> HBM (pageable and pinned) + one pass.  Reported in DESIGN.md
section 5; bench.py's `value` starts with the inputs resident in HBM."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config("c2")
F = 256
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
eng.run(masks="rle"); torch.cuda.synchronize()
host_bytes = hb.raw.nbytes + hb.rle_counts.nbytes + hb.cams.nbytes + hb.sweep_xf.nbytes + hb.lane.nbytes
for label in ("pageable", "pageable"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.upload(hb); eng.run(masks="rle"); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: upload (incl. device allocations) + pass {dt * 1e3:.2f} ms -> {F / dt:.0f} frames/s; {host_bytes / 1e6:.0f} MB of inputs")
# pinned staging + asynchronous copies of the large arrays
raw_pin = torch.from_numpy(hb.raw).pin_memory(); rle_pin = torch.from_numpy(hb.rle_counts.view(np.int32)).pin_memory()
dev_raw = torch.empty_like(raw_pin, device="cuda"); dev_rle = torch.empty_like(rle_pin, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev_raw.copy_(raw_pin, non_blocking=True); dev_rle.copy_(rle_pin, non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.run(masks="rle"); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pinned: copy {1e3 * (t1 - t0):.2f} ms ({(raw_pin.nbytes + rle_pin.nbytes) / (t1 - t0) / 1e9:.1f} GB/s), pass {1e3 * (t2 - t1):.2f} ms "
          f"-> {F / (t2 - t0):.0f} frames/s serial, {F / max(t1 - t0, t2 - t1):.0f} frames/s when the copy of batch i+1 overlaps pass i")
# host-fed LiftPipeline: every step uploads a packed host batch (all inputs), runs the pass and downloads the results of
# the batch two steps back -- what the nuScenes entry point does with batches it has read and packed
if True:
    hbs = [lifting.pack_frames([syn.make_frame(cfg, (k + 1) * F + i) for i in range(F)], lanes, [0] * F) for k in range(3)]
    pipe = lifting.LiftPipeline(depth=3)
    pending = []
    for k in range(6):                          # warm-up: allocator pools, pinned blocks
        if len(pending) == pipe.depth:
            pipe.collect(pending.pop(0), full=False)
        pending.append(pipe.submit(hbs[k % 3], "rle"))
    while pending:
        pipe.collect(pending.pop(0), full=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 18
    for k in range(K):
        if len(pending) == pipe.depth:
            pipe.collect(pending.pop(0), full=False)
        pending.append(pipe.submit(hbs[k % 3], "rle"))
    while pending:
        pipe.collect(pending.pop(0), full=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"LiftPipeline host-fed, {dt * 1e3:.2f} ms per batch incl. upload and download -> {F / dt:.0f} frames/s")
