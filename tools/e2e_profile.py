#!/usr/bin/env python3
"""Runs ON THE GPU BOX: where the host side of the nuScenes entry point spends its time -- native tables, table walk, mask files,
sweep files, upload -- on a synthetic C2-shaped dataset (one batch = 8 scenes x 32 frames)."""
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cm3d_amd import lifting, nusc_io, reader, synthetic as syn

cfg = syn.config("c2")
root = tempfile.mkdtemp(prefix="cm3d_prof_")
dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(root, cfg, n_scenes=32, frames_per_scene=32, lane_points=50000, pool=64)
rd = reader.Reader(0)
t0 = time.perf_counter(); nt = reader.Tables(rd, dataroot, "v1.0-synth"); print(f"tables {1e3 * (time.perf_counter() - t0):.1f} ms ({rd.threads} threads)")
classes = lifting.ClassTable.nuscenes()
for rep in range(3):
    for b0 in range(0, 32, 8):
        bn = names[b0:b0 + 8]
        t = [time.perf_counter()]
        man = nt.manifest(bn, mask_dir, 1, 1.0, classes.names); t.append(time.perf_counter())
        c, ro, fmo, wh = man.load_masks(); t.append(time.perf_counter())
        raw, off = man.load_sweeps(5); t.append(time.perf_counter())
        d = torch.from_numpy(np.asarray(raw)).to("cuda:0", non_blocking=True); torch.cuda.synchronize(); t.append(time.perf_counter())
        if rep == 2:
            print(f"batch {b0 // 8}: manifest {1e3 * (t[1] - t[0]):.1f}  masks {1e3 * (t[2] - t[1]):.1f}  sweeps {1e3 * (t[3] - t[2]):.1f} ms "
                  f"({raw.nbytes / 1e6:.0f} MB, {raw.nbytes / (t[3] - t[2]) / 1e9:.1f} GB/s)  upload {1e3 * (t[4] - t[3]):.1f} ms")
# the same batches through _native_batch + LiftEngine.upload / run, timed on the host
from cm3d_amd import pipeline_nuscenes as pn
eng = lifting.LiftEngine()
lane_cache = {}
import cProfile, pstats
for rep in range(3):
    for b0 in range(0, 32, 8):
        t = [time.perf_counter()]
        hb, rows = pn._native_batch(nt, names[b0:b0 + 8], mask_dir, 1, 1.0, classes, False, lane_cache); t.append(time.perf_counter())
        if rep == 2 and b0 == 8:
            h = torch.from_numpy(np.ascontiguousarray(hb.raw))
            torch.cuda.synchronize(); q0 = time.perf_counter(); x = h.to("cuda:0", non_blocking=h.is_pinned()); q1 = time.perf_counter(); torch.cuda.synchronize(); q2 = time.perf_counter()
            print("raw pinned?", h.is_pinned(), type(hb.raw), hb.raw.flags["C_CONTIGUOUS"], f".to call {1e3 * (q1 - q0):.2f} ms, done {1e3 * (q2 - q0):.2f} ms")
            del x
            pr = cProfile.Profile(); pr.enable()
        eng.upload(hb); t.append(time.perf_counter())
        if rep == 2 and b0 == 8:
            pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
        eng.run(masks="rle"); t.append(time.perf_counter())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        rec = lifting.kept_box_records(eng.b, np.zeros((hb.n_frames, 2))); torch.cuda.synchronize(); t.append(time.perf_counter())
        if rep == 2:
            print(f"batch {b0 // 8}: native batch {1e3 * (t[1] - t[0]):.1f}  upload (host) {1e3 * (t[2] - t[1]):.1f}  run (host) {1e3 * (t[3] - t[2]):.1f}  "
                  f"wait {1e3 * (t[4] - t[3]):.1f}  records {1e3 * (t[5] - t[4]):.1f} ms")
import shutil
shutil.rmtree(root, ignore_errors=True)
