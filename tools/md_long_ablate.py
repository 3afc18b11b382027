#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so): what the medoid
stage of a batch of LONG lists (c1, c5) spends where -- the stage timed alone with parts of it switched off (results wrong by
construction; the engine's other stages have run before and are not repeated).
Usage: python tools/md_long_ablate.py [config=c1] [frames=256]"""
import ctypes as C
import os
import pickle
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cm3d_amd import _lib, lifting, synthetic as syn

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cache = os.path.join(os.environ.get("CM3D_BENCH_CACHE", "/tmp"), f"mdl_{name}_{F}.pkl")
lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=7, extent=260.0)]
if os.path.exists(cache):
    hb = pickle.load(open(cache, "rb"))
else:
    cfg = syn.config(name)
    frames = []
    for i in range(F):
        frames.append(syn.make_frame(cfg, i))
        if (i + 1) % 32 == 0:
            print(f"generating frame {i + 1} / {F}", file=sys.stderr, flush=True)
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    pickle.dump(hb, open(cache, "wb"), protocol=4)
eng = lifting.LiftEngine()
eng.upload(hb)
L = _lib.lib()
L.cm3d_md_diag_set.argtypes = [C.c_int]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
names = {128: "rows staged once per tile", 256: "no second pass", 512: "first-pass tiles only",
         1024: "exact tiles only"}
for flags in (0, 256, 256 + 512, 256 + 1024, 256 + 512 + 128, 0):
    L.cm3d_md_diag_set(flags)
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.stage_medoid(st)
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts[2:])
    print(f"flags {flags:5d}  {ms[len(ms) // 2] * 1e3:8.1f} us   " + (" + ".join(v for k, v in names.items() if flags & k) or "full stage"), flush=True)
import numpy as np
L.cm3d_md_diag_set(2048)
eng.stage_medoid(st)
torch.cuda.synchronize()
pos = eng.b.medoid_pos.cpu().numpy()
ho = eng.b.hit_off.cpu().numpy()
Ms = np.diff(ho)[:len(pos)]
lm = (Ms > 256) & (Ms < 100000) & (Ms.max() > 448)
C = pos[lm]
print(f"long masks {lm.sum()} of {len(pos)}; M: median {int(np.median(Ms[lm]))} max {Ms[lm].max()}; candidates per long mask: min {C.min()} median {int(np.median(C))} "
      f"mean {C.mean():.1f} p90 {int(np.percentile(C, 90))} max {C.max()}; share with more than 16: {np.mean(C > 16):.3f}, more than 64: {np.mean(C > 64):.3f}; "
      f"sum over masks of ceil(C/16) * M = {int((np.ceil(C / 16) * Ms[lm]).sum())} rows walked, longest walk {int((np.ceil(C / 16) * Ms[lm]).max())}")
L.cm3d_md_diag_set(0)
