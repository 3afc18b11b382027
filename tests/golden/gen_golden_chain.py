#!/usr/bin/env python3
"""Round-4 pins of the last reference-own steps that were only held through the oracle (missing #2 and #5), generated
like gen_golden.py by RUNNING THE REFERENCE'S OWN CODE in the build container (where /root/reference exists); never runs on the GPU
box.  Writes

  g2e_sweep_loop.npz   the sweep loop of src/nuscenes/2d_to_3d.py:437-465 re-run statement by statement on the imported
                       LidarPointCloud.from_file / .rotate / .translate -- temp .bin files of one c1-shaped frame (3 sweeps) per
                       magnitude (ego pose 0 m, 1.7 km, 4 km from the map origin), the |x| < sqrt(2.3) and |y| < sqrt(2.3) filter of
                       :442-445, torch.hstack of the sweeps -- and, on THAT cloud, the per-mask loop body (:543-617, gen_golden.
                       reference_mask_body).  Per magnitude: sha256 of the cloud's bytes, the rows the filter dropped (per sweep),
                       every 97th point's bits, and the index lists.  The rotation matrices / translations handed to rotate /
                       translate are the frame's float32 records (what torch.from_numpy(Quaternion(q).rotation_matrix).to(float32)
                       yields; pyquaternion itself is third-party and stays unpinned).
  g7r_tiny_scene.json  one tiny scene (2 frames) chained through the imported get_medoid -> lane_yaws_distances_and_coords ->
                       get_detection_name -> get_shape_prior -> push_centroid -> circle_nms, with the driver code between them -- the
                       running id_offset / centroid_ids bookkeeping of :410,511,663,730-744 and the per-sample NMS loop of :844-924 --
                       restated once, here.  Output: the reference's final_predictions["results"] for the scene.
                       (Quaternion(matrix=Rz(yaw)) is gen_golden.pyquaternion_from_rz, the restated third-party piece.)
Usage: python tests/golden/gen_golden_chain.py   (from the repo root)
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as gg  # noqa: E402
from cm3d_amd import lifting, nusc_io, rle as rlemod, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

G2E = [dict(config="c1", index=31, over=dict(n_masks=24, ego_magnitude=mag)) for mag in (0.0, 1700.0, 4000.0)]
MIN_DIST = 2.3           # reference :348


def reference_sweep_loop(pcd, bin_paths, sweep_xf):
    """2d_to_3d.py:437-465 for one frame: files -> aggr_pc_points (4, N) float32, and per sweep the rows the filter dropped."""
    DEVICE = "cpu"
    aggr_set, dropped = [], []
    for path, xf in zip(bin_paths, sweep_xf):
        pc = pcd.LidarPointCloud.from_file(path, DEVICE)                                         # :439
        lidar_points = pc.points                                                                 # :441
        mask = torch.ones(lidar_points.shape[1]).to(device=DEVICE)                               # :442
        mask = torch.logical_and(mask, torch.abs(lidar_points[0, :]) < np.sqrt(MIN_DIST))       # :443
        mask = torch.logical_and(mask, torch.abs(lidar_points[1, :]) < np.sqrt(MIN_DIST))       # :444
        lidar_points = lidar_points[:, ~mask]                                                    # :445
        pc = pcd.LidarPointCloud(lidar_points)                                                   # :446
        xf = np.asarray(xf, np.float32)
        pc.rotate(torch.from_numpy(xf[0:9].reshape(3, 3).copy()).to(device=DEVICE, dtype=torch.float32))       # :451
        pc.translate(torch.from_numpy(xf[9:12].copy()).to(device=DEVICE, dtype=torch.float32))                  # :452
        pc.rotate(torch.from_numpy(xf[12:21].reshape(3, 3).copy()).to(device=DEVICE, dtype=torch.float32))      # :456
        pc.translate(torch.from_numpy(xf[21:24].copy()).to(device=DEVICE, dtype=torch.float32))                 # :457
        aggr_set.append(pc.points)                                                               # :459
        dropped.append(np.flatnonzero(mask.numpy()).astype(np.int32))
    aggr_pc_points = torch.hstack(tuple([p for p in aggr_set]))                                  # :465
    return aggr_pc_points.numpy(), dropped


def g2e(pcd, report):
    out = {}
    for k, spec in enumerate(G2E):
        cfg = syn.config(spec["config"], **spec["over"])
        f = syn.make_frame(cfg, spec["index"])
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, r in enumerate(f.sweeps_raw):
                p = os.path.join(td, f"s{i}.pcd.bin")
                np.ascontiguousarray(r, np.float32).tofile(p)
                paths.append(p)
            P4N, dropped = reference_sweep_loop(pcd, paths, f.sweep_xf)
        P = np.ascontiguousarray(P4N.T)                                    # (N, 4), the layout of oracle.sweep_prep's output
        # the oracle on the same inputs
        Po = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(f.sweeps_raw, f.sweep_xf)], 0)
        n_mis_cloud = int(Po.shape != P.shape) or int((Po.view(np.uint32) != P.view(np.uint32)).sum())
        lists, n_mis = [], 0
        for r, c in zip(f.rles, f.cam_nums):
            m = rlemod.counts_to_dense(rlemod.string_to_counts(r["counts"]), f.width, f.height)
            er = orc.erode3x3(m)
            tp, _ = gg.reference_mask_body(pcd, P, f.cams[c], er)
            n_mis += int(not np.array_equal(tp, orc.points_in_mask(P, f.cams[c], er)))
            lists.append(tp)
        tag = f"m{k}"
        out[f"{tag}_spec"] = json.dumps(spec)
        out[f"{tag}_sha256"] = hashlib.sha256(P.tobytes()).hexdigest()
        out[f"{tag}_n_points"] = np.int64(P.shape[0])
        out[f"{tag}_dropped"] = np.concatenate(dropped)
        out[f"{tag}_dropped_off"] = np.concatenate([[0], np.cumsum([d.size for d in dropped])]).astype(np.int32)
        out[f"{tag}_sample"] = P[::97].copy()
        out[f"{tag}_idx"] = np.concatenate(lists).astype(np.int32)
        out[f"{tag}_idx_off"] = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.int32)
        report[f"G2e magnitude {spec['over']['ego_magnitude']:g} m: points / dropped rows / in-mask points"] = \
            [int(P.shape[0]), int(sum(d.size for d in dropped)), int(sum(l.size for l in lists))]
        report[f"G2e magnitude {spec['over']['ego_magnitude']:g} m: cloud words that differ (oracle vs reference sweep loop)"] = n_mis_cloud
        report[f"G2e magnitude {spec['over']['ego_magnitude']:g} m: index-list mismatches (oracle vs reference body on the reference's cloud)"] = n_mis
    np.savez_compressed(os.path.join(HERE, "g2e_sweep_loop.npz"), **out)


class _Quat:
    """The two things the chained functions do with a pyquaternion.Quaternion: list(q) = [w, x, y, z]."""

    def __init__(self, wxyz):
        self.q = [float(v) for v in wxyz]

    def __iter__(self):
        return iter(self.q)


def g7r(pcd, ref, report):
    """The tiny scene of G7 (same dataset writer, same seed; tests.helpers.g7r_scene) through the reference's own functions."""
    from tests.helpers import g7r_scene
    with tempfile.TemporaryDirectory() as td:
        frames, lane_pt_list = g7r_scene(td)         # G7's tiny scene + three detections per frame listed twice (NMS must act)
    shape_priors = lifting.SHAPE_PRIORS_CHATGPT                     # cfg/shape_priors_chatgpt.json (:384-385)
    # ---- stage 1 (:415-694): per frame, per mask; id_offset numbers every mask of the scene in file order (:410,511)
    id_offset = -1
    centroid_ids, all_centroids_list = [], []
    for f in frames:
        P = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(f.sweeps_raw, f.sweep_xf)], 0)
        aggr = torch.from_numpy(np.ascontiguousarray(P.T))          # (the sweep loop itself is G2e's subject)
        for r, c in zip(f.rles, f.cam_nums):
            id_offset += 1                                           # :511
            m = rlemod.counts_to_dense(rlemod.string_to_counts(r["counts"]), f.width, f.height)
            tp, _ = gg.reference_mask_body(pcd, P, f.cams[c], orc.erode3x3(m))
            if tp.size == 0:                                         # :626
                continue
            pts_in_mask = aggr[:3, torch.from_numpy(tp.astype(np.int64))]          # :620,645
            medoid_idx = ref.get_medoid(pts_in_mask)                 # :646
            all_centroids_list.append(pts_in_mask[:, medoid_idx])    # :647-662
            centroid_ids.append(id_offset)                           # :663
    all_centroids = torch.stack(all_centroids_list)                  # :697
    all_centroids = torch.squeeze(all_centroids)                     # :698
    yaw_list, min_distance_list, _ = ref.lane_yaws_distances_and_coords(all_centroids, lane_pt_list)      # :702
    # ---- stage 2 (:728-822): the JSONs walked again with the same counter
    predictions = {"results": {}}
    id_offset = -1
    for f in frames:
        predictions["results"][f.token] = []                         # :733
        for label, score, c in zip(f.labels, f.scores, f.cam_nums):
            id_offset += 1                                           # :735
            if id_offset not in centroid_ids:                        # :736
                continue
            idx = centroid_ids.index(id_offset)                      # :739
            detection_name = ref.get_detection_name(label)           # :743
            centroid = np.squeeze(np.array(all_centroids[idx, :]))   # :744
            lane_yaw = yaw_list[idx]                                 # :750
            extents = ref.get_shape_prior(shape_priors, detection_name)        # :760
            if detection_name in ["car", "truck", "bus", "construction_vehicle", "trailer", "barrier"]:      # :763
                align_mat = np.eye(3)                                # :788
                align_mat[0:2, 0:2] = [[np.cos(lane_yaw), -np.sin(lane_yaw)], [np.sin(lane_yaw), np.cos(lane_yaw)]]     # :789
                poserecord = {"translation": [float(v) for v in f.ego_xyz]}    # :793-795: the keyframe's LIDAR_TOP ego pose
                q = _Quat(gg.pyquaternion_from_rz(lane_yaw))         # Quaternion(matrix=align_mat), restated third-party
                pushed_centroid = ref.push_centroid(centroid, extents, q, poserecord)      # :796
            else:
                q = _Quat([1.0, 0.0, 0.0, 0.0])                      # Quaternion(matrix=eye(3))
                pushed_centroid = centroid                           # :802
            predictions["results"][f.token].append({                 # :808-817
                "sample_token": f.token,
                "translation": [float(i) for i in pushed_centroid],
                "size": list(extents),
                "rotation": list(q),
                "velocity": [0, 0],
                "detection_name": detection_name,
                "detection_score": score,
                "attribute_name": lifting.ATTRIBUTE_NAMES[detection_name],
            })
    # ---- NMS per sample (:844-924)
    final = {}
    for sample in predictions["results"]:
        final[sample] = []
        dets, det_labels, boxes = [], [], predictions["results"][sample]
        for box_dict in boxes:
            centroid = box_dict["translation"]
            dets.append(np.array([centroid[0], centroid[1], box_dict["detection_score"]]))      # :880
            det_labels.append(box_dict["detection_name"])
        dets = np.array(dets)
        if len(det_labels) > 0:
            keep_indices = list(ref.circle_nms(dets, det_labels, lifting.THRESHS_BY_LABEL))     # :893
        else:
            continue                                                 # :896
        for k, box_dict in enumerate(boxes):                         # :901-924: the kept ones, in their original order
            if k in keep_indices:
                final[sample].append(box_dict)
    json.dump({"results": final, "masks_in_scene": id_offset + 1, "masks_with_points": len(centroid_ids)},
              open(os.path.join(HERE, "g7r_tiny_scene.json"), "w"))
    report["G7r reference-chained tiny scene: masks / with points / boxes before NMS / after"] = \
        [id_offset + 1, len(centroid_ids), int(sum(len(v) for v in predictions["results"].values())), int(sum(len(v) for v in final.values()))]
    # the oracle pipeline on the same scene (what G7 holds): how far apart
    from tests.helpers import oracle_results
    res7 = oracle_results(orc, frames, [lane_pt_list], [0] * len(frames))
    worst, same_sets = 0.0, True
    for tok, want in final.items():
        got = res7.get(tok, [])
        same_sets &= len(got) == len(want) and all(a["detection_name"] == b["detection_name"] and a["detection_score"] == b["detection_score"]
                                                   for a, b in zip(got, want))
        if len(got) == len(want):
            for a, b in zip(got, want):
                worst = max(worst, float(np.abs(np.array(a["translation"]) - np.array(b["translation"])).max()),
                            float(np.abs(np.array(a["rotation"]) - np.array(b["rotation"])).max()))
    report["G7r oracle pipeline keeps the same boxes in the same order"] = bool(same_sets)
    report["G7r max |oracle - reference chain| over translation and rotation"] = worst


def main():
    pcd, ref = gg._load_reference()
    report = json.load(open(os.path.join(HERE, "gen_report.json")))
    g2e(pcd, report)
    g7r(pcd, ref, report)
    json.dump(report, open(os.path.join(HERE, "gen_report.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in report.items() if k.startswith(("G2e", "G7r"))}, indent=1))


if __name__ == "__main__":
    main()
