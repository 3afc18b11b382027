#!/usr/bin/env python3
"""Runs ON THE GPU BOX under `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) with the diagnostic library: calibrates the counter
on a launch whose HBM bytes are known.  With the camera loop switched off (diag flag 4) k_project_hits reads every raw row once
(4 x stride x rows bytes, through the same 12-byte-per-row requests as the product) and writes one all-zero hit word per row
and plane -- nothing else of any size.  Sequence of k_project_hits dispatches in this process: 2 full launches (warm-up),
then 3 with flag 4, then 3 full ones; tools/pmc_fetch_calibration.sh reads the counter of each group.
usage: [CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so] tools/fetch_calibration.py [config frames]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import _lib, lifting, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = syn.config(name)
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
L = _lib.lib()
L.cm3d_diag_set.argtypes = [C.c_int]
st = torch.cuda.current_stream().cuda_stream
eng.stage_begin(st)
eng.stage_masks(st, "rle")
torch.cuda.synchronize()
for flags, reps in ((0, 2), (4, 3), (0, 3)):
    L.cm3d_diag_set(flags)
    for _ in range(reps):
        eng.stage_sweep_project(st)
        torch.cuda.synchronize()
print(f"rows {hb.n_raw_rows} stride {hb.raw_stride} planes {eng.b.planes}: raw rows {4 * hb.raw_stride * hb.n_raw_rows} B, "
      f"hit words {4 * eng.b.planes * hb.n_raw_rows} B")
