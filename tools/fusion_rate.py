#!/usr/bin/env python3
"""Rate of the SAM3D fusion matching at nuScenes-val scale: 6019 samples, ~25 lifted boxes against ~45 SAM3D
boxes each, one cm3d_bev_match call with the records resident in HBM; the CPU oracle on a sample of the same input beside it.
Prints one JSON line."""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cm3d_amd import _lib, ops  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (CPU baseline leg only)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6019
rng = np.random.default_rng(1)


def boxes(n, centre, spread):
    c = centre + rng.uniform(-spread, spread, (n, 2))
    return np.stack([c[:, 0], c[:, 1], rng.uniform(-1, 1, n), rng.uniform(1.5, 5.5, n), rng.uniform(0.8, 2.5, n), rng.uniform(1, 2, n),
                     rng.uniform(-np.pi, np.pi, n)], 1)


preds, gts = [], []
for f in range(S):
    P = int(rng.integers(5, 45))
    p = boxes(P, rng.uniform(-1, 1, 2) * (700.0, 1500.0), 40.0)
    g = p.copy()
    g[:, :2] += rng.normal(0, 0.5, (P, 2)); g[:, 6] += rng.normal(0, 0.2, P)
    g = np.concatenate([g[rng.random(P) < 0.7], boxes(int(rng.integers(10, 50)), p[0, :2], 40.0)])
    preds.append(p); gts.append(g)
pr, gr = [ops.match_records(b) for b in preds], [ops.match_records(b) for b in gts]
np_, ng = np.array([len(r) for r in pr]), np.array([len(r) for r in gr])
dev = torch.device("cuda")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
p_off, g_off = np.concatenate([[0], np.cumsum(np_)]).astype(np.int32), np.concatenate([[0], np.cumsum(ng)]).astype(np.int32)
pair_off = np.concatenate([[0], np.cumsum(np_ * ng)]).astype(np.int64)
d_p, d_g, d_po, d_go, d_pair = t(np.concatenate(pr)), t(np.concatenate(gr)), t(p_off), t(g_off), t(pair_off)
n_pred, n_gt, total = int(p_off[-1]), int(g_off[-1]), int(pair_off[-1])
pm, gm = torch.empty(n_pred, dtype=torch.int32, device=dev), torch.empty(n_gt, dtype=torch.int32, device=dev)
iou = torch.empty(n_pred, dtype=torch.float64, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
L = _lib.lib()
ws = torch.empty(L.cm3d_bev_match_workspace_bytes(total), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    _lib.check(L.cm3d_bev_match(d_p.data_ptr(), d_po.data_ptr(), n_pred, d_g.data_ptr(), d_go.data_ptr(), n_gt, d_pair.data_ptr(), S, total, 0.2,
                                pm.data_ptr(), gm.data_ptr(), iou.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), st), "cm3d_bev_match")


for _ in range(3):
    run()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
n_cpu = S
t0 = time.perf_counter()
for f in range(n_cpu):
    orc.bev_match(pr[f], gr[f], 0.2)
dt_cpu = time.perf_counter() - t0
print(json.dumps({"metric": "fusion matching samples/s", "samples": S, "pairs": total, "matches": int((pm >= 0).sum().item()),
                  "ms_per_call": round(dt * 1e3, 3), "value": round(S / dt, 1), "pairs_per_s": round(total / dt, 1),
                  "cpu_baseline": {"value": round(n_cpu / dt_cpu, 1), "unit": "samples/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_cpu} samples, {dt_cpu:.2f} s"}}))
