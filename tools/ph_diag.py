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
for flags in (0, 32, 2, 4, 8, 12, 0):
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
    ph = ["setup", "raw wait+xform", "wedge tests", "projection", "mask loop", "result stores", "-"]
    print(f"stamps (flags {flags}): {int(v[7])} waves, {tot / max(v[7], 1):.0f} cycles per wave")
    for k in range(7):
        print(f"   {ph[k]:24s} {100 * v[k] / tot:5.1f} %   {v[k] / max(v[7], 1):9.0f} cycles/wave")
    out[f"stamps_{flags}"] = {ph[k]: round(100 * v[k] / tot, 1) for k in range(7)}
L.cm3d_diag_set(64)
eng.stage_begin(st)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
eng.stage_sweep_project(st)
b.record()
torch.cuda.synchronize()
print(f"counting launch: {a.elapsed_time(b) * 1e3:.1f} us")
buf = (C.c_ulonglong * 8)()
L.cm3d_diag_read_counts(buf)
v = [int(x) for x in buf]
print(f"counts: {v[0]} wave-chunks; per chunk: {v[1] / v[0]:.3f} cameras behind the wedge, {v[2] / v[0]:.3f} behind the pre-test, "
      f"{v[3] / v[0]:.3f} with a point in the image, {v[4] / v[0]:.3f} mask batches, {v[5] / v[0]:.3f} masks")
out["counts"] = v
L.cm3d_diag_set(128)
for _ in range(3):
    eng.stage_begin(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.stage_sweep_project(st)
    b.record()
    torch.cuda.synchronize()
NW = 16384
wv = (C.c_ulonglong * (3 * NW))()
L.cm3d_diag_read_waves(wv, NW)
wi = np.array(list(wv), np.uint64).reshape(NW, 3)
live = wi[:, 1] > 0
wi = wi[live]
w = wi[:, :2].astype(np.float64)
hw, xcc = (wi[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), (wi[:, 2] >> np.uint64(32)).astype(np.int64) & 15
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7)       # cu_id, se_id, sh_id
simd = (hw >> 4) & 3
life = w[:, 1] - w[:, 0]
print(f"timed launch {a.elapsed_time(b) * 1e3:.1f} us; {len(w)} waves; lifetime ticks: min {life.min():.0f} p10 {np.percentile(life, 10):.0f} "
      f"median {np.median(life):.0f} mean {life.mean():.0f} p90 {np.percentile(life, 90):.0f} max {life.max():.0f}")
# per workgroup (4 consecutive waves): spread of the four lifetimes; start skew inside the launch (clocks of different XCDs
# are not comparable, so only the spread of starts modulo a workgroup is shown)
g = life[: len(life) // 4 * 4].reshape(-1, 4)
print(f"inside a workgroup: mean (max - min) of the four wave lifetimes = {np.mean(g.max(1) - g.min(1)):.0f} ticks; mean of max = {g.max(1).mean():.0f}")
fr = life[: len(life) // 16 * 16].reshape(-1, 16)       # 16 waves (4 workgroups) per frame at the default grid
fm = fr.mean(1)
print(f"per frame (mean of its 16 waves): min {fm.min():.0f} p10 {np.percentile(fm, 10):.0f} median {np.median(fm):.0f} p90 {np.percentile(fm, 90):.0f} max {fm.max():.0f}; "
      f"mean spread inside a frame {np.mean(fr.max(1) - fr.min(1)):.0f}")
hc = eng.b.hit_count.cpu().numpy()
mo = hb.mask_off
hits_f = np.array([hc[mo[i]:mo[i + 1]].sum() for i in range(F)], np.float64)
print(f"correlation of a frame's wave lifetime with its hit total: {np.corrcoef(fm, hits_f[:len(fm)])[0, 1]:.3f}")
lf = w[:, 1] - w[:, 0]
for x in range(8):
    sel = xcc == x
    if sel.any():
        st0 = w[sel, 0].min()
        print(f"  XCD {x}: {sel.sum():4d} waves, {len(set(cu[sel]))} CUs; lifetime mean {lf[sel].mean():.0f} max {lf[sel].max():.0f}; "
              f"starts spread {w[sel, 0].max() - st0:.0f}; last end - first start {w[sel, 1].max() - st0:.0f}")
key = xcc * 1024 + cu * 4 + simd
ks, cnts = np.unique(key, return_counts=True)
print(f"waves per SIMD: min {cnts.min()} max {cnts.max()} over {len(ks)} SIMDs")
busy = np.array([w[key == k, 1].max() - w[key == k, 0].min() for k in ks])
print(f"per SIMD busy span: min {busy.min():.0f} median {np.median(busy):.0f} max {busy.max():.0f}")
span = busy.max()
res = np.array([lf[key == k].sum() for k in ks]) / span
print(f"mean resident waves per SIMD over the longest SIMD span: {res.mean():.2f} (min {res.min():.2f} max {res.max():.2f}); "
      f"SIMD busy span / longest: mean {busy.mean() / span:.3f}")
ck = xcc * 256 + cu
cs = np.unique(ck)
cbusy = np.array([w[ck == k, 1].max() - w[ck == k, 0].min() for k in cs])
cn = np.array([(ck == k).sum() for k in cs])
print(f"per CU: waves min {cn.min()} median {int(np.median(cn))} max {cn.max()}; span min {cbusy.min():.0f} median {np.median(cbusy):.0f} max {cbusy.max():.0f}; "
      f"corr(waves, span) {np.corrcoef(cn, cbusy)[0, 1]:.2f}")
out["wave_life"] = dict(min=float(life.min()), median=float(np.median(life)), mean=float(life.mean()), max=float(life.max()))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/ph_diag_{name}.json", "w"), indent=1)
