#!/usr/bin/env python3
"""Host-side cost of enqueueing one pass (LiftEngine.run) against the GPU time of the pass: the pass is GPU-bound
as long as the first stays well below the second."""
import time
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config("c2")
F = 256
frames = [syn.make_frame(cfg, i) for i in range(F)]
import sys
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
for _ in range(3):
    eng.run(masks="rle")
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    eng.run(masks="rle")
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / n:.1f} us per pass, wall {1e6 * (t2 - t0) / n:.1f} us per pass")

for label, with_events in (("plain", False), ("with 2 event records per pass", True), ("plain again", False)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if with_events:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        eng.run(masks="rle")
        if with_events:
            b.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: enqueue {1e6 * (t1 - t0) / n:.1f} us, wall {1e6 * (t2 - t0) / n:.1f} us per pass")

st = torch.cuda.current_stream().cuda_stream
calls = {"sweeps": eng.stage_sweeps, "masks": lambda s: eng.stage_masks(s, "rle"), "project": eng.stage_project,
         "compact": eng.stage_compact, "medoid": eng.stage_medoid, "lanes": lambda s: (eng.wait_lane_grid(), eng.stage_lanes(s)),
         "boxes": eng.stage_boxes}
ev = {k: [] for k in calls}
for _ in range(10):
    eng.stage_begin(st)
    for k, fn in calls.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(st); b.record()
        ev[k].append((a, b))
torch.cuda.synchronize()
print({k: round(sum(a.elapsed_time(b) for a, b in v) / len(v), 4) for k, v in ev.items()})
print("status", eng.check_status(), "max hits", int(eng.b.hit_count.max().item()))
