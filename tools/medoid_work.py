#!/usr/bin/env python3
"""Prints the distribution of in-mask point counts of the default bench batch and the medoid work
it implies (64-column tiles x rows), to tell a work-bound k_medoid_tiles from a tail-bound one."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn

cfg = syn.config(sys.argv[1] if len(sys.argv) > 1 else "c2")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = [syn.make_frame(cfg, i) for i in range(F)]
lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 50000, seed=1, extent=260.0)]
hb = lifting.pack_frames(frames, lanes, [0] * F)
eng = lifting.LiftEngine()
eng.upload(hb)
eng.run(masks="rle")
torch.cuda.synchronize()
got = eng.download()
M = np.diff(got["hit_off"]).astype(np.int64)
tiles = (M + 63) // 64
print("masks", M.size, "nonempty", int((M > 0).sum()), "sum M", int(M.sum()), "max", int(M.max()))
print("percentiles 50/90/99:", np.percentile(M[M > 0], [50, 90, 99]))
print("tiles", int(tiles.sum()), "wave-rows (sum tiles*M)", int((tiles * M).sum()), "pairs M^2", int((M * M).sum()))
print("M<=25 (direct):", int(((M > 0) & (M <= 25)).sum()))
print("ten longest lists:", sorted(M.tolist())[-10:])
for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 4096), (4097, 10**9)):
    sel = (M >= lo) & (M <= hi)
    print(f"M in [{lo},{hi}]: {int(sel.sum())} masks, wave-rows {int((tiles * M)[sel].sum())}")
