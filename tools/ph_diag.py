#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so):
times the fused projection launch of one resident batch under the ablation switches of CM3D_DIAG builds and prints
the per-phase s_memtime shares.  Results of ablated launches are wrong by construction; only times are read."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
L.cm3d_diag_read.argtypes = [C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
names = {2: "no mask loop", 4: "no camera loop", 8: "synthetic rows (no raw loads)", 32: "no approximate pre-test"}
out = {}
for flags in (0, 32, 2, 4, 8, 12, 0, 32):
    L.cm3d_diag_set(flags)
    ts = []
    for _ in range(12):
        eng.stage_begin(st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.stage_sweep_project(st)
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts[2:])
    label = " + ".join(v for k, v in names.items() if flags & k) or "full kernel"
    out[f"{flags}: {label}"] = round(ms[len(ms) // 2] * 1e3, 1)
    print(f"flags {flags:2d}  {ms[len(ms) // 2] * 1e3:8.1f} us   {label}", flush=True)
for flags in (16, 28):
    L.cm3d_diag_set(flags)
    eng.stage_begin(st)
    eng.stage_sweep_project(st)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    L.cm3d_diag_read(buf)
    v = np.array(list(buf), np.float64)
    tot = v[:7].sum()
    ph = ["setup", "raw wait+xform", "cone tests", "projection", "mask loop", "result stores", "-"]
    print(f"stamps (flags {flags}): {int(v[7])} waves, {tot / max(v[7], 1):.0f} cycles per wave")
    for k in range(7):
        print(f"   {ph[k]:24s} {100 * v[k] / tot:5.1f} %   {v[k] / max(v[7], 1):9.0f} cycles/wave")
    out[f"stamps_{flags}"] = {ph[k]: round(100 * v[k] / tot, 1) for k in range(7)}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/ph_diag_{name}.json", "w"), indent=1)
