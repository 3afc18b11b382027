#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so):
median time of the fused projection launch of one resident batch for each diag flag word given on the command line
(ablated launches produce wrong results by construction; only times are read).
  python tools/ph_ablate.py c2 256 0 256 512 768 0"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import _lib, lifting, synthetic as syn

name, F = sys.argv[1], int(sys.argv[2])
flag_list = [int(x) for x in sys.argv[3:]] or [0]
cfg = syn.config(name)
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
L = _lib.lib()
L.cm3d_diag_set.argtypes = [C.c_int]
st = torch.cuda.current_stream().cuda_stream
eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
for flags in flag_list:
    L.cm3d_diag_set(flags)
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
    print(f"flags {flags:5d}  median {ms[len(ms) // 2] * 1e3:8.1f} us   min {ms[0] * 1e3:8.1f} us", flush=True)
