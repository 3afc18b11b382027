#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so):
where the time of k_medoid_tiles goes -- per wave (= per tile) its start, the end of its first staging and its end
(s_memtime, 100 MHz), its SIMD and its list length."""
import ctypes as C
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
L.cm3d_md_diag_set.argtypes = [C.c_int]
L.cm3d_md_diag_read_waves.argtypes = [C.c_void_p, C.c_int]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
names = {2: "no root", 4: "no domain test / zero route", 8: "no LDS reads", 16: "one add per step"}
for flags in (0, 4, 2, 6, 8, 16, 22, 30, 0):
    L.cm3d_md_diag_set(flags)
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.stage_medoid(st)
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts[2:])
    print(f"flags {flags:2d}  {ms[len(ms) // 2] * 1e3:8.1f} us   " + (" + ".join(v for k, v in names.items() if flags & k) or "full kernel"), flush=True)
L.cm3d_md_diag_set(1)
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.stage_medoid(st)
    b.record()
    torch.cuda.synchronize()
ck = (C.c_ulonglong * 4)()
L.cm3d_md_diag_read_clock.argtypes = [C.c_void_p]
L.cm3d_md_diag_read_clock(ck)
print(f"tile 0's wave: {ck[2] - ck[0]} s_memtime ticks in {(ck[3] - ck[1]) * 10} ns of the 100 MHz clock -> {(ck[2] - ck[0]) / max(1, (ck[3] - ck[1]) * 10):.3f} ticks per ns")
NW = 65536
wv = (C.c_ulonglong * (5 * NW))()
L.cm3d_md_diag_read_waves(wv, NW)
wi = np.array(list(wv), np.uint64).reshape(NW, 5)
wi = wi[wi[:, 2] > 0]
t = wi[:, :3].astype(np.float64)
hw, xcc = (wi[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64), (wi[:, 3] >> np.uint64(32)).astype(np.int64) & 15
M = wi[:, 4].astype(np.int64)
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7)
simd = (hw >> 4) & 3
print(f"medoid stage {a.elapsed_time(b) * 1e3:.1f} us; {len(t)} tiles; M: min {M.min()} median {int(np.median(M))} mean {M.mean():.0f} max {M.max()}")
life, front, comp = t[:, 2] - t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
tick = 0.01   # printed unit: 100 s_memtime ticks
for nm, v in (("lifetime", life), ("start -> first rows staged", front), ("staged -> end", comp)):
    print(f"  {nm:28s} us: min {v.min() * tick:6.2f} p10 {np.percentile(v, 10) * tick:6.2f} median {np.median(v) * tick:6.2f} mean {v.mean() * tick:6.2f} "
          f"p90 {np.percentile(v, 90) * tick:6.2f} max {v.max() * tick:6.2f}")
print(f"  compute time per row (staged -> end) / M, ns: median {np.median(comp / M) * 10:.1f} p10 {np.percentile(comp / M, 10) * 10:.1f} p90 {np.percentile(comp / M, 90) * 10:.1f}")
for x in range(8):
    sel = xcc == x
    if sel.any():
        s0 = t[sel, 0].min()
        print(f"  XCD {x}: {sel.sum():4d} waves on {len(set(cu[sel]))} CUs; starts spread {(t[sel, 0].max() - s0) * tick:6.2f} us; first start -> last end {(t[sel, 2].max() - s0) * tick:6.2f} us; "
              f"sum of rows {M[sel].sum()}")
key = xcc * 1024 + cu * 4 + simd
ks, cnts = np.unique(key, return_counts=True)
rows = np.array([M[key == k].sum() for k in ks])
span = np.array([t[key == k, 2].max() - t[key == k, 0].min() for k in ks]) * tick
print(f"per SIMD ({len(ks)} used): waves min {cnts.min()} median {int(np.median(cnts))} max {cnts.max()}; rows min {rows.min()} median {int(np.median(rows))} max {rows.max()}; "
      f"span us min {span.min():.2f} median {np.median(span):.2f} max {span.max():.2f}; corr(rows, span) {np.corrcoef(rows, span)[0, 1]:.2f}")
worst = ks[np.argsort(span)[-5:]]
for k in worst:
    sel = key == k
    s0 = t[sel, 0].min()
    print(f"  slow SIMD {k}: " + "; ".join(f"M {m} [{(a0 - s0) * tick:.1f} {(a1 - s0) * tick:.1f} {(a2 - s0) * tick:.1f}]" for m, a0, a1, a2 in zip(M[sel], t[sel, 0], t[sel, 1], t[sel, 2])))
# how evenly the launch ends: when each SIMD's last tile is done, relative to the first start anywhere, and how many tiles it held on average
t0 = t[:, 0].min()
last = np.array([t[key == k, 2].max() - t0 for k in ks]) * tick
busy = np.array([(t[key == k, 2] - t[key == k, 0]).sum() for k in ks]) * tick
print(f"launch span {(t[:, 2].max() - t0) * tick:.1f} us; a SIMD's last tile ends at: min {last.min():.1f} p10 {np.percentile(last, 10):.1f} median {np.median(last):.1f} "
      f"p90 {np.percentile(last, 90):.1f} max {last.max():.1f} us; tiles resident per SIMD (sum of lifetimes / span): median {np.median(busy / ((t[:, 2].max() - t0) * tick)):.2f}")
longm = M > 256
for nm, sel in (("first-pass tiles (M > 256)", longm), ("exact tiles", ~longm)):
    if sel.any():
        lt = (t[sel, 2] - t[sel, 0]) * tick
        print(f"  {nm}: {sel.sum()} tiles, {M[sel].sum()} rows; lifetime per row ns: median {np.median(lt / M[sel]) * 1e3:.1f}; starts at us: median {np.median(t[sel, 0] - t0) * tick:.1f} "
              f"p90 {np.percentile(t[sel, 0] - t0, 90) * tick:.1f} max {(t[sel, 0].max() - t0) * tick:.1f}")
