#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so):
median time of the projection launch (k_project_q, quad layout) of one resident batch with the chunk loop cut after
stage 0 rows + results + draws | 1 + sweep transform | 2 + view wedges | 3 + approximate pre-test | 4 + exact chain | 5 everything.
Cut launches produce wrong results by construction; only times are read.
  python tools/pq_stages.py c2 256"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import _lib, lifting, synthetic as syn

name, F = sys.argv[1], int(sys.argv[2])
cfg = syn.config(name)
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F, layout="quads")
eng = lifting.LiftEngine()
eng.upload(hb)
L = _lib.lib()
L.cm3d_diag_pq_stage.argtypes = [C.c_int]
st = torch.cuda.current_stream().cuda_stream
eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
for stage in (99,):
    L.cm3d_diag_pq_stage(stage)
    ts = []
    for _ in range(14):
        eng.stage_begin(st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.stage_sweep_project(st)
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts[2:])
    print(f"stage {stage:3d}  median {ms[len(ms) // 2] * 1e3:8.1f} us   min {ms[0] * 1e3:8.1f} us", flush=True)
# single intervals (two s_memtime per chunk): 101 the wait at the end of an iteration (draw + the rows requested before it),
# 103 the gather batches, 104 the camera loop (wedges .. last camera)
L.cm3d_diag_set.argtypes = [C.c_int]
L.cm3d_diag_read.argtypes = [C.c_void_p]
L.cm3d_diag_read_waves.argtypes = [C.c_void_p, C.c_int]
for which, what in ((101, "end-of-iteration wait"), (103, "gather batches"), (104, "camera loop"), (105, "rows -> global coordinates, drops"), (106, "results out"), (107, "prefetch + draw issued")):
    for extra in (0,):
        L.cm3d_diag_set(0)
        L.cm3d_diag_pq_stage(which + extra)
        eng.stage_begin(st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.stage_sweep_project(st)
        b.record()
        torch.cuda.synchronize()
        NW = 3072
        wv = (C.c_ulonglong * (3 * NW))()
        L.cm3d_diag_read_waves(wv, NW)
        import numpy as np
        w = np.array(list(wv), np.float64).reshape(NW, 3)
        w = w[w[:, 0] > 0]
        print(f"interval {which} ({what}){' static draws' if extra else ''}: launch {a.elapsed_time(b) * 1e3:.1f} us; {len(w)} waves; per wave {w[:, 0].mean():.0f} cycles alive, "
              f"{w[:, 1].mean():.0f} in the interval ({100 * w[:, 1].sum() / w[:, 0].sum():.1f} %), {w[:, 2].mean():.1f} intervals, {w[:, 1].sum() / max(w[:, 2].sum(), 1):.0f} cycles each")
