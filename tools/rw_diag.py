#!/usr/bin/env python3
"""Runs ON THE GPU BOX with the diagnostic library (make -C cm3d_amd/csrc diag; CM3D_LIB=cm3d_amd/libcm3d_hip_diag.so):
per-wave (= per-mask) timeline of k_rle_erode_pack_wave -- start, end (s_memtime), SIMD, run count."""
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
L.cm3d_rw_diag_set.argtypes = [C.c_int]
L.cm3d_rw_diag_read_waves.argtypes = [C.c_void_p, C.c_int]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.run(masks="rle")
torch.cuda.synchronize()
L.cm3d_rw_diag_set(1)
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.stage_masks(st, "rle")
    b.record()
    torch.cuda.synchronize()
NW = min(32768, hb.n_masks)
wv = (C.c_ulonglong * (4 * NW))()
L.cm3d_rw_diag_read_waves(wv, NW)
wi = np.array(list(wv), np.uint64).reshape(NW, 4)
wi = wi[wi[:, 1] > 0]
t = wi[:, :2].astype(np.float64)
hw, xcc = (wi[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), (wi[:, 2] >> np.uint64(32)).astype(np.int64) & 15
n = wi[:, 3].astype(np.int64)
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7)
simd = (hw >> 4) & 3
life = t[:, 1] - t[:, 0]
print(f"mask stage {a.elapsed_time(b) * 1e3:.1f} us; {len(t)} masks; runs per mask: min {n.min()} median {int(np.median(n))} mean {n.mean():.0f} p90 {int(np.percentile(n, 90))} max {n.max()}")
print(f"wave lifetime, kilo-ticks of s_memtime: min {life.min() / 1e3:.1f} p10 {np.percentile(life, 10) / 1e3:.1f} median {np.median(life) / 1e3:.1f} mean {life.mean() / 1e3:.1f} "
      f"p90 {np.percentile(life, 90) / 1e3:.1f} max {life.max() / 1e3:.1f}; corr(runs, lifetime) {np.corrcoef(n, life)[0, 1]:.2f}")
for lo, hi in ((0, 50), (50, 150), (150, 300), (300, 600), (600, 10**9)):
    sel = (n >= lo) & (n < hi)
    if sel.any():
        print(f"  runs {lo:4d}..{hi if hi < 10**9 else 'inf'}: {sel.sum():5d} masks, lifetime mean {life[sel].mean() / 1e3:6.1f} k, ticks per run {np.mean(life[sel] / np.maximum(n[sel], 1)):7.1f}")
key = xcc * 1024 + cu * 4 + simd
ks, cnts = np.unique(key, return_counts=True)
span = np.array([t[key == k, 1].max() - t[key == k, 0].min() for k in ks])
busy = np.array([life[key == k].sum() for k in ks])
print(f"per SIMD ({len(ks)} used): waves min {cnts.min()} median {int(np.median(cnts))} max {cnts.max()}; span k-ticks min {span.min() / 1e3:.1f} median {np.median(span) / 1e3:.1f} max {span.max() / 1e3:.1f}; "
      f"mean resident waves over its span {np.mean(busy / span):.2f}")
for x in range(8):
    sel = xcc == x
    if sel.any():
        s0 = t[sel, 0].min()
        print(f"  XCD {x}: {sel.sum():5d} waves; starts spread {(t[sel, 0].max() - s0) / 1e3:7.1f} k; first start -> last end {(t[sel, 1].max() - s0) / 1e3:7.1f} k")
worst = ks[np.argsort(span)[-3:]]
for k in worst:
    sel = key == k
    s0 = t[sel, 0].min()
    print(f"  slow SIMD {k}: " + "; ".join(f"n {m} [{(a0 - s0) / 1e3:.1f} {(a1 - s0) / 1e3:.1f}]" for m, a0, a1 in zip(n[sel], t[sel, 0], t[sel, 1])))
