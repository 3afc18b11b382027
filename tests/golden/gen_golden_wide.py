#!/usr/bin/env python3
"""Wider reference-run pins (round 2), generated like gen_golden.py by RUNNING THE REFERENCE'S OWN CODE in the build
container (where /root/reference exists); never runs on the GPU box.  Writes

  g2b_c1_frame.npz    G2 on a frame of the reference's own configuration (3 sweeps x 34.7 k points, 6 cameras, masks of
                      1024x576 at ratio 0.64, 24 masks): the per-mask loop body of src/nuscenes/2d_to_3d.py:543-617 re-run
                      on the imported LidarPointCloud / view_points -> one index list per mask.  The inputs are the
                      committed generator's frame (config + index below) and are pinned by a checksum.
  g2c_c2_frame.npz    the same on a frame of the headline configuration (BASELINE C2: 35 k points, 6 cameras, 20 masks of
                      1600x900 at ratio 1.0).
  g3b_medoid_lists.npz  G3 on REAL in-mask lists: the reference's get_medoid (:116-119, torch.cdist + sum + argmin) on the
                      global-frame points of >= 300 masks of c1-shaped frames, a third of them with duplicated points;
                      per list the reference's index, the oracle's index and the oracle's best/second-best margin.
Usage: python tests/golden/gen_golden_wide.py   (from the repo root)
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as gg  # noqa: E402
from cm3d_amd import rle as rlemod, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

G2B = dict(config="c1", index=11, over=dict(n_masks=24))
G2C = dict(config="c2", index=5, over=dict())
G3B_FRAMES = [dict(config="c1", index=100 + i, over=dict(n_masks=24)) for i in range(16)]


def frame_cloud(spec):
    """The frame of `spec` and its aggregated global-frame cloud (the reference's aggr_pc_points, :437-465)."""
    cfg = syn.config(spec["config"], **spec["over"])
    f = syn.make_frame(cfg, spec["index"])
    P = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(f.sweeps_raw, f.sweep_xf)], 0)
    return cfg, f, P


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def frame_checksum(f, P):
    return checksum(P, f.cams, np.array(f.cam_nums, np.int32), *[rlemod.string_to_counts(r["counts"]) for r in f.rles])


def main():
    pcd, ref = gg._load_reference()
    report = json.load(open(os.path.join(HERE, "gen_report.json")))

    # ---------------- G2b, G2c
    for tag, spec, fn, shape in (("G2b c1-shaped frame", G2B, "g2b_c1_frame.npz", (1024, 576, 0.64)),
                                 ("G2c c2-shaped frame", G2C, "g2c_c2_frame.npz", (1600, 900, 1.0))):
        cfg, f, P = frame_cloud(spec)
        assert (cfg.width, cfg.height, cfg.ratio) == shape
        lists, n_mis = [], 0
        for r, c in zip(f.rles, f.cam_nums):
            m = rlemod.counts_to_dense(rlemod.string_to_counts(r["counts"]), f.width, f.height)
            er = orc.erode3x3(m)
            tp, _ = gg.reference_mask_body(pcd, P, f.cams[c], er)
            n_mis += int(not np.array_equal(tp, orc.points_in_mask(P, f.cams[c], er)))
            lists.append(tp)
        np.savez_compressed(os.path.join(HERE, fn), spec=json.dumps(spec), sha256=frame_checksum(f, P),
                            idx=np.concatenate(lists).astype(np.int32),
                            idx_off=np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.int32), n_points=np.int64(P.shape[0]))
        report[f"{tag}: masks / points / in-mask points"] = [len(lists), int(P.shape[0]), int(sum(l.size for l in lists))]
        report[f"{tag[:3]} index-list mismatches (oracle vs reference body)"] = n_mis

    # ---------------- G3b
    rng = np.random.default_rng(20240202)
    pts_all, off, ref_idx, orc_idx, margin, dup = [], [0], [], [], [], []
    for spec in G3B_FRAMES:
        cfg, f, P = frame_cloud(spec)
        masks = [rlemod.counts_to_dense(rlemod.string_to_counts(r["counts"]), f.width, f.height) for r in f.rles]
        for m, c in zip(masks, f.cam_nums):
            il = orc.points_in_mask(P, f.cams[c], orc.erode3x3(m))
            if il.size == 0:
                continue
            p = P[il, :3].astype(np.float32)
            if len(pts_all) % 3 == 0 and p.shape[0] >= 4:
                # duplicated rows (a sweep listed twice, a point returned twice): exact ties between columns, first index wins
                extra = p[rng.choice(p.shape[0], max(1, p.shape[0] // 10), replace=False)]
                p = np.concatenate([p, extra], 0)[rng.permutation(p.shape[0] + extra.shape[0])]
            j_ref = int(ref.get_medoid(torch.from_numpy(np.ascontiguousarray(p.T))))
            P4 = np.concatenate([p, np.zeros((p.shape[0], 1), np.float32)], 1)
            j_orc, cs = orc.medoid(P4, np.arange(p.shape[0]), want_colsum=True)
            srt = np.sort(cs)
            mg = float((srt[1] - srt[0]) / srt[0]) if p.shape[0] > 1 and srt[0] > 0 else float("inf")
            pts_all.append(p); off.append(off[-1] + p.shape[0])
            ref_idx.append(j_ref); orc_idx.append(int(j_orc)); margin.append(mg)
            dup.append(int(np.unique(p, axis=0).shape[0] < p.shape[0]))
    ref_idx, orc_idx, margin, dup = np.array(ref_idx), np.array(orc_idx), np.array(margin), np.array(dup)
    agree = ref_idx == orc_idx
    # a disagreement can also be a tie in VALUE: the two picks are then the same point twice (a duplicated row)
    same_point = np.array([np.array_equal(pts_all[k][ref_idx[k]], pts_all[k][orc_idx[k]]) for k in range(len(pts_all))])
    np.savez_compressed(os.path.join(HERE, "g3b_medoid_lists.npz"), pts=np.concatenate(pts_all, 0), off=np.array(off, np.int64),
                        ref_index=ref_idx.astype(np.int32), oracle_index=orc_idx.astype(np.int32), rel_margin=margin, has_duplicates=dup.astype(np.int8))
    lens = np.diff(off)
    report["G3b lists / with duplicated points / longest"] = [int(len(pts_all)), int(dup.sum()), int(lens.max())]
    report["G3b reference get_medoid == oracle (index)"] = f"{int(agree.sum())} of {len(agree)}"
    report["G3b disagreements that pick the same coordinates (duplicated rows)"] = int((~agree & same_point).sum())
    report["G3b disagreements: list length, rel margin"] = [[int(lens[k]), float(margin[k])] for k in np.flatnonzero(~agree)]
    json.dump(report, open(os.path.join(HERE, "gen_report.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in report.items() if k.startswith(("G2b", "G2c", "G3b"))}, indent=1))


if __name__ == "__main__":
    main()
