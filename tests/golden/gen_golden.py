#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json by RUNNING THE REFERENCE'S OWN HELPERS in the build
container (where /root/reference exists).  Never runs on the GPU box; the fixtures it writes are
data (inputs + the outputs the reference produced) and are committed.

What is imported from the reference (read-only, with inert stand-in modules for third-party
imports that are not installed here and that these helpers never call -- cv2, pycocotools,
pyquaternion, shapely, nuscenes, ...):
    src/nuscenes/utils/pcd.py   : LidarPointCloud.translate / .rotate, view_points
    src/nuscenes/2d_to_3d.py    : get_medoid, push_centroid, circle_nms,
                                  lane_yaws_distances_and_coords, get_detection_name, get_shape_prior
The per-mask loop body (2d_to_3d.py:553-620) is inline script code; G2 re-runs it here statement by
statement on top of the imported LidarPointCloud / view_points with the same torch calls.

Usage: python tests/golden/gen_golden.py   (from the repo root)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/nuscenes"

from cm3d_amd import geometry as geo, rle as rlemod, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Stub(self.__name__ + "." + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


def _load_reference():
    absent = ["cv2", "torchvision", "pycocotools", "hdbscan", "pyquaternion", "groundingdino.datasets.transforms",
              "groundingdino.models", "groundingdino.util.slconfig", "groundingdino.util.utils", "segment_anything",
              "shapely.geometry", "nuscenes.nuscenes", "nuscenes.utils.data_classes", "nuscenes.utils.geometry_utils",
              "nuscenes.map_expansion.map_api", "nuscenes.map_expansion.arcline_path_utils", "nuscenes.map_expansion.bitmap",
              "nuscenes.utils.splits", "numba"]
    for name in absent:
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            n = ".".join(parts[:i])
            try:
                if n not in sys.modules:
                    importlib.import_module(n)
            except Exception:
                sys.modules[n] = _Stub(n)

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    pcd = load(os.path.join(REF, "utils/pcd.py"), "utils.pcd")
    utils = types.ModuleType("utils")
    utils.pcd = pcd
    sys.modules["utils"] = utils
    ref = load(os.path.join(REF, "2d_to_3d.py"), "ref_2d_to_3d")
    ref.timer = {"closest lane": 0}       # module global the function accumulates into (2d_to_3d.py:300)
    return pcd, ref


def _load_waymo_reference():
    """src/waymo/2d_to_3d.py with inert stand-ins for tensorflow / waymo_open_dataset / ... (never called by
    the two helpers used here)."""
    for name in ["tensorflow", "tensorflow.compat.v1", "waymo_open_dataset", "waymo_open_dataset.utils",
                 "waymo_open_dataset.utils.range_image_utils", "waymo_open_dataset.utils.transform_utils",
                 "waymo_open_dataset.utils.frame_utils", "waymo_open_dataset.dataset_pb2", "waymo_open_dataset.label_pb2",
                 "waymo_open_dataset.protos", "waymo_open_dataset.protos.metrics_pb2", "pycocotools.mask", "open3d", "trimesh",
                 "cfg", "cfg.prompt_cfg"]:
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            n = ".".join(parts[:i])
            if n not in sys.modules:
                try:
                    importlib.import_module(n)
                except Exception:
                    sys.modules[n] = _Stub(n)
    wdir = "/root/reference/src/waymo"
    spec = importlib.util.spec_from_file_location("utils.pcd", os.path.join(wdir, "utils/pcd.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["utils.pcd"] = m
    spec.loader.exec_module(m)
    sys.modules["utils"].pcd = m
    spec = importlib.util.spec_from_file_location("ref_waymo_2d_to_3d", os.path.join(wdir, "2d_to_3d.py"))
    w = importlib.util.module_from_spec(spec)
    sys.modules["ref_waymo_2d_to_3d"] = w
    spec.loader.exec_module(w)
    return w


def _load_kitti_reference():
    """src/kitti/kitti_utils.py (Calibration, inverse_rigid_trans) with inert stand-ins for cv2 etc."""
    for name in ["cv2", "mayavi", "mayavi.mlab"]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Stub(name)
    spec = importlib.util.spec_from_file_location("ref_kitti_utils", "/root/reference/src/kitti/kitti_utils.py")
    k = importlib.util.module_from_spec(spec)
    sys.modules["ref_kitti_utils"] = k
    spec.loader.exec_module(k)
    return k


def reference_mask_body(pcd, pts, cam, eroded_hw, min_dist=2.3):
    """2d_to_3d.py:543-617 on CPU tensors; cam = our float32 camera record (the tensors the
    reference builds at :570-577,:585-587).  Returns track_points (ascending point indices)."""
    DEVICE = "cpu"
    aggr_pc_points = torch.from_numpy(np.ascontiguousarray(pts.T))            # (4,N)
    maskarr_1 = np.asarray(eroded_hw)[:, :].astype(bool)
    maskarr_1 = torch.transpose(torch.from_numpy(maskarr_1).to(device=DEVICE, dtype=bool), 1, 0)   # :544
    track_points = np.array(range(aggr_pc_points.shape[1]))                   # :548
    cam_pc = pcd.LidarPointCloud(torch.clone(aggr_pc_points))                 # :553
    stages, flags = int(cam[54]), int(cam[55])
    for s in range(stages):
        assert flags & (1 << (2 * s)) and not flags & (2 << (2 * s)), "nuScenes / Waymo stages translate, then rotate"
        cam_pc.translate(torch.from_numpy(cam[15 * s:15 * s + 3].copy()))     # :570 / :576
        cam_pc.rotate(torch.from_numpy(cam[15 * s + 3:15 * s + 12].reshape(3, 3).copy()))   # :571 / :577
    depths = cam_pc.points[2, :]                                              # :581
    camera_intrinsic = torch.from_numpy(cam[45:54].reshape(3, 3).copy())
    points, point_depths = pcd.view_points(cam_pc.points[:3, :], camera_intrinsic, normalize=True, device=DEVICE)  # :590
    image_mask = maskarr_1
    masked_pixels = (image_mask == 1)
    points_within_image = torch.logical_and(torch.logical_and(torch.logical_and(torch.logical_and(
        depths > min_dist, points[0] > 0), points[0] < image_mask.shape[0] - 1), points[1] > 0),
        points[1] < image_mask.shape[1] - 1)                                  # :597-603
    floored_points = torch.floor(points[:, points_within_image]).to(dtype=int)   # :605
    track_points = track_points[points_within_image.cpu()]                    # :606
    points_within_mask = torch.logical_and(floored_points, masked_pixels[floored_points[0], floored_points[1]])   # :608-611
    indices_within_mask = torch.where(torch.logical_and(torch.logical_and(
        points_within_mask[0, :], points_within_mask[1, :]), points_within_mask[2, :]))[0]   # :613
    track_points = track_points[indices_within_mask.cpu()]                    # :617
    uvd = torch.stack([points[0], points[1], depths]).T.contiguous().numpy()
    return np.asarray(track_points, np.int32).reshape(-1), uvd


def pyquaternion_from_rz(yaw_f32):
    """What `Quaternion(matrix=align_mat)` holds for align_mat = Rz(lane_yaw) built at
    2d_to_3d.py:788-789 (np.cos/np.sin of a float32 are float32).  pyquaternion 0.9.9 is not
    installed here; this is its trace method on M^T, restated."""
    c, s = float(np.cos(np.float32(yaw_f32))), float(np.sin(np.float32(yaw_f32)))
    if c < -c:
        t = 1.0 - c - c + 1.0
        q = np.array([2 * s, 0.0, 0.0, t]) * (0.5 / np.sqrt(t))
    else:
        t = 1.0 + c + c + 1.0
        q = np.array([t, 0.0, 0.0, 2 * s]) * (0.5 / np.sqrt(t))
    return q


def main():
    pcd, ref = _load_reference()
    rng = np.random.default_rng(20240101)
    report = {}

    # ---------------- G1: translate / rotate / view_points, bit-exact float32
    cfg = syn.config("tiny")
    fr = syn.make_frame(cfg, 3)
    pts = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(fr.sweeps_raw, fr.sweep_xf)], 0)
    pts = pts[rng.choice(pts.shape[0], 4096, replace=False)]
    g1_uvd = []
    for c in range(fr.cams.shape[0]):
        _, uvd = reference_mask_body(pcd, pts, fr.cams[c], np.ones((cfg.height, cfg.width), np.uint8))
        g1_uvd.append(uvd)
        mine = orc.project_points(pts, fr.cams[c])
        both_nan = np.isnan(mine) & np.isnan(uvd)
        report[f"G1 cam{c} mismatches"] = int((~both_nan & (mine.view(np.uint32) != uvd.view(np.uint32))).sum())
    # a single-stage (Waymo-style) camera too
    cam1 = geo.single_stage_cam_record(fr.cams[0][0:3], fr.cams[0][3:12], fr.cams[0][45:54])
    _, uvd1 = reference_mask_body(pcd, pts, cam1, np.ones((cfg.height, cfg.width), np.uint8))
    np.savez_compressed(os.path.join(HERE, "g1_project.npz"), pts=pts, cams=fr.cams, uvd=np.stack(g1_uvd), cam1=cam1, uvd1=uvd1)

    # ---------------- G2: index lists of whole frames + crafted boundary points
    g2 = {}
    n_mis = 0
    for k, idx in enumerate([0, 1]):
        f = syn.make_frame(cfg, idx)
        P = np.concatenate([orc.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(f.sweeps_raw, f.sweep_xf)], 0)
        # crafted points: pixel coordinates near 0, 1, W-1, H-1 and depth near min_dist, back-projected to global
        extra = []
        for c in range(f.cams.shape[0]):
            cam = f.cams[c].astype(np.float64)
            (t1, R1, _), (t2, R2, _), K = geo.cam_stage(cam, 0), geo.cam_stage(cam, 1), geo.cam_K(cam)
            for _ in range(160):
                u = rng.choice([0.0, 1.0, cfg.width - 1.0, rng.uniform(0, cfg.width)]) + rng.normal() * 1e-3
                v = rng.choice([0.0, 1.0, cfg.height - 1.0, rng.uniform(0, cfg.height)]) + rng.normal() * 1e-3
                z = rng.choice([2.3, rng.uniform(2.0, 40.0)]) + rng.normal() * 1e-6
                pc = np.array([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z])
                pg = R1.T @ (R2.T @ pc - t2) - t1
                extra.append([pg[0], pg[1], pg[2], 1.0])
        P = np.concatenate([P, np.array(extra, np.float32)], 0)
        masks = [rlemod.counts_to_dense(rlemod.string_to_counts(r["counts"]), f.width, f.height) for r in f.rles]
        masks[0] = np.ones_like(masks[0])                      # a full-frame mask: every in-image point
        lists = []
        for m, c in zip(masks, f.cam_nums):
            er = orc.erode3x3(m)
            tp, _ = reference_mask_body(pcd, P, f.cams[c], er)
            mine = orc.points_in_mask(P, f.cams[c], er)
            n_mis += int(not np.array_equal(tp, mine))
            lists.append(tp)
        g2[f"pts{k}"] = P
        g2[f"cams{k}"] = f.cams
        g2[f"cam_nums{k}"] = np.array(f.cam_nums, np.int32)
        g2[f"rle_counts{k}"] = np.concatenate([rlemod.dense_to_counts(m) for m in masks])
        g2[f"rle_off{k}"] = np.concatenate([[0], np.cumsum([rlemod.dense_to_counts(m).size for m in masks])]).astype(np.int32)
        g2[f"idx{k}"] = np.concatenate(lists) if lists else np.zeros(0, np.int32)
        g2[f"idx_off{k}"] = np.concatenate([[0], np.cumsum([l.size for l in lists])]).astype(np.int32)
    g2["wh"] = np.array([cfg.width, cfg.height], np.int32)
    report["G2 index-list mismatches (oracle vs reference body)"] = n_mis
    report["G2 total in-mask points"] = int(g2["idx0"].size + g2["idx1"].size)
    np.savez_compressed(os.path.join(HERE, "g2_index_lists.npz"), **g2)

    # ---------------- G3: get_medoid
    g3 = {}
    cases = []
    for M in [1, 2, 3, 17, 25, 26, 27, 64, 100, 300, 1000, 2000]:
        for tag, off in [("local", np.zeros(3)), ("global", np.array([612.3, 1587.9, 1.7]))]:
            p = (rng.normal(size=(M, 3)) * [4.0, 4.0, 0.6] + off).astype(np.float32)
            j_ref = int(ref.get_medoid(torch.from_numpy(p.T.copy())))
            P4 = np.concatenate([p, np.zeros((M, 1), np.float32)], 1)
            j_orc, cs = orc.medoid(P4, np.arange(M), want_colsum=True)
            srt = np.sort(cs)
            margin = float((srt[1] - srt[0]) / srt[0]) if M > 1 and srt[0] > 0 else float("inf")
            name = f"M{M}_{tag}"
            g3[name] = p
            cases.append({"name": name, "M": M, "ref_index": j_ref, "oracle_index": j_orc, "rel_margin": margin})
    report["G3 medoid index mismatches"] = int(sum(c["ref_index"] != c["oracle_index"] for c in cases))
    np.savez_compressed(os.path.join(HERE, "g3_medoid.npz"), **g3)
    json.dump(cases, open(os.path.join(HERE, "g3_medoid.json"), "w"), indent=1)

    # ---------------- G4: push_centroid
    g4 = []
    priors = orc.PRIORS_WLH
    worst = 0.0
    for i in range(240):
        cls = int(rng.integers(0, len(orc.CLASSES)))
        yaw = np.float32(rng.uniform(-np.pi, np.pi)) if i % 7 else np.float32([0.0, np.pi / 2, -np.pi / 2, np.pi][i % 4])
        ego = np.array([600.0 + rng.uniform(-50, 50), 1600.0 + rng.uniform(-50, 50), rng.uniform(0, 2)])
        d = rng.uniform(3, 60)
        a = rng.uniform(-np.pi, np.pi)
        cen = (ego + [d * np.cos(a), d * np.sin(a), rng.uniform(-1, 2)]).astype(np.float32)
        if i % 31 == 0:
            cen[1] = np.float32(ego[1])      # ey ~ 0
        q = pyquaternion_from_rz(yaw)
        pushed = ref.push_centroid(cen.copy(), list(priors[cls]), list(q), {"translation": list(ego)})
        t, qo = orc.box_assemble(cen, priors[cls], yaw, ego, True)
        worst = max(worst, float(np.abs(t - pushed).max()), float(np.abs(qo - q).max()))
        g4.append({"centroid": [float(v) for v in cen], "class": cls, "yaw": float(yaw), "ego": list(map(float, ego)),
                   "pushed": [float(v) for v in pushed], "quat_wxyz": [float(v) for v in q]})
    report["G4 push_centroid max |oracle - reference|"] = worst
    json.dump(g4, open(os.path.join(HERE, "g4_push_centroid.json"), "w"))

    # ---------------- G5: circle_nms (distinct scores -> no tie ambiguity) + pinned tie fixture
    g5 = []
    names = orc.CLASSES
    thr = {n: float(t) for n, t in zip(names, orc.NMS_THR)}
    bad = 0
    for i in range(40):
        n = int(rng.integers(1, 60))
        centers = rng.uniform(0, 12, size=(max(1, n // 3), 2))
        xy = centers[rng.integers(0, centers.shape[0], n)] + rng.normal(scale=0.6, size=(n, 2))
        scores = rng.permutation(np.arange(1, n + 1) / (n + 1.0))                 # all distinct
        labels = [names[int(k)] for k in rng.integers(0, 4 if i % 2 else len(names), n)]
        dets = np.concatenate([xy, scores[:, None]], 1)
        keep = sorted(int(k) for k in ref.circle_nms(dets, labels, thr))
        mine = np.flatnonzero(orc.circle_nms(xy[:, 0], xy[:, 1], scores, [names.index(l) for l in labels], orc.NMS_THR)).tolist()
        bad += int(keep != mine)
        g5.append({"xy": xy.tolist(), "scores": scores.tolist(), "labels": labels, "keep": keep})
    report["G5 circle_nms mismatches"] = bad
    # ties: scores rounded to 2 decimals; expectation = the pinned tie-break (descending score, then descending index)
    n = 30
    xy = rng.normal(scale=1.0, size=(n, 2))
    scores = np.round(rng.uniform(0.3, 0.5, n), 2)
    labels = ["car"] * n
    keep = np.flatnonzero(orc.circle_nms(xy[:, 0], xy[:, 1], scores, [0] * n, orc.NMS_THR)).tolist()
    json.dump({"reference_cases": g5, "tie_case_pinned": {"xy": xy.tolist(), "scores": scores.tolist(), "labels": labels, "keep": keep}},
              open(os.path.join(HERE, "g5_circle_nms.json"), "w"))

    # ---------------- G6: lane_yaws_distances_and_coords
    lane = syn.make_lane_table([600.0, 1600.0], 6000, seed=5)
    cent = np.stack([600 + rng.uniform(-150, 150, 300), 1600 + rng.uniform(-150, 150, 300), rng.uniform(0, 3, 300)], 1).astype(np.float32)
    # duplicates in the lane table exercise the first-minimum rule
    lane[100] = lane[50]
    lane[3000] = lane[2999]
    yaws, dists, coords = ref.lane_yaws_distances_and_coords(cent.tolist(), lane.tolist())
    j, d = orc.lane_nn(cent, lane)
    lane32 = lane.astype(np.float32)
    report["G6 lane NN mismatches"] = int((lane32[j, 2] != yaws).sum() + (d != dists).sum())
    np.savez_compressed(os.path.join(HERE, "g6_lane_nn.npz"), lane=lane, centroids=cent, yaws=np.asarray(yaws), dists=np.asarray(dists),
                        coords=np.asarray(coords), oracle_idx=j)

    # ---------------- G8: Waymo deltas (src/waymo/2d_to_3d.py: push_centroid(ego_frame=True), get_yaws_from_lane_coords)
    wref = _load_waymo_reference()
    g8 = {"push": [], "lanes": []}
    worst8 = 0.0
    for i in range(120):
        cls = int(rng.integers(0, len(orc.CLASSES)))
        yaw = np.float32(rng.uniform(-np.pi, np.pi))
        d, a = rng.uniform(3, 60), rng.uniform(-np.pi, np.pi)
        cen = np.array([d * np.cos(a), d * np.sin(a), rng.uniform(-1, 2)])
        q = pyquaternion_from_rz(yaw)
        pushed = wref.push_centroid(cen.copy(), list(priors[cls]), list(q), ego_frame=True)
        # the oracle takes a global centroid and the inverse pose: identity pose makes them the same frame
        t, _ = orc.box_assemble_waymo(cen.astype(np.float32), np.eye(4, dtype=np.float32).reshape(16), priors[cls], yaw, True)
        pushed32 = wref.push_centroid(cen.astype(np.float32).astype(np.float64), list(priors[cls]), list(q), ego_frame=True)
        worst8 = max(worst8, float(np.abs(t - pushed32).max()))
        g8["push"].append({"centroid": [float(v) for v in cen.astype(np.float32)], "class": cls, "yaw": float(yaw),
                           "pushed": [float(v) for v in pushed32]})
    from types import SimpleNamespace as NS
    for n in [1, 2, 5, 40]:
        poly = np.cumsum(rng.normal(size=(n, 3)), 0) + [5000.0, -3000.0, 20.0]
        out = wref.get_yaws_from_lane_coords([NS(x=float(p_[0]), y=float(p_[1]), z=float(p_[2])) for p_ in poly])
        g8["lanes"].append({"polyline": poly.tolist(), "xyyaw": np.asarray(out).tolist()})
    report["G8 waymo push_centroid max |oracle - reference|"] = worst8
    json.dump(g8, open(os.path.join(HERE, "g8_waymo.json"), "w"))

    # ---------------- G9: KITTI chain (src/kitti/kitti_utils.py Calibration, src/kitti/2d_to_3d.py:1066-1073,1238-1256)
    kref = _load_kitti_reference()
    import tempfile
    from cm3d_amd import kitti as kt
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    R0 = geo.rot_z(0.002) @ geo.quat_to_rotmat([1, 0.004, -0.003, 0.001])
    Rv = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float) @ geo.quat_to_rotmat([1, 0.003, 0.002, -0.004])
    tv = np.array([-0.004069766, -0.07631618, -0.2717806])
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
        fh.write("P2: " + " ".join(repr(float(v)) for v in P2.reshape(-1)) + "\n")
        fh.write("R0_rect: " + " ".join(repr(float(v)) for v in R0.reshape(-1)) + "\n")
        fh.write("Tr_velo_to_cam: " + " ".join(repr(float(v)) for v in np.hstack([Rv, tv[:, None]]).reshape(-1)) + "\n")
        calib_path = fh.name
    rc = kref.Calibration(calib_path)
    velo = (rng.uniform(-1, 1, (3000, 4)) * [40, 25, 2, 1] + [22, 0, -0.6, 0]).astype(np.float32)
    ref_pts = rc.project_velo_to_ref(torch.from_numpy(velo[:, :3]))                       # :1070-1073
    depths_ = rc.project_ref_to_velo(torch.clone(ref_pts))                                # :1238
    rect = rc.project_velo_to_rect(depths_)                                               # :1239-1240
    Kk = torch.Tensor([[rc.f_u, 0, rc.c_u], [0, rc.f_v, rc.c_v], [0, 0, 1]]).to(dtype=torch.float32) * 0.8366
    Kk[2, 2] = 1
    pts_img, _ = pcd.view_points(rect.T[:3, :], Kk, normalize=True, device="cpu")
    mine_cal = kt.Calibration(calib_path)
    rec = mine_cal.cam_record()
    xf = mine_cal.sweep_xf()
    mine_ref = orc.sweep_prep(velo, xf[0:9], xf[9:12], xf[12:21], xf[21:24], np.float32(0.0))
    uvd = orc.project_points(mine_ref, rec)
    ref_uvd = np.stack([pts_img[0].numpy(), pts_img[1].numpy(), rect[:, 2].numpy()], 1)
    both_nan = np.isnan(uvd) & np.isnan(ref_uvd)
    report["G9 kitti velo->ref mismatches"] = int((mine_ref[:, :3].view(np.uint32) != ref_pts.numpy().view(np.uint32)).sum())
    report["G9 kitti chain u,v,depth mismatches"] = int((~both_nan & (uvd.view(np.uint32) != ref_uvd.view(np.uint32))).sum())
    np.savez_compressed(os.path.join(HERE, "g9_kitti.npz"), velo=velo, P2=P2, R0=R0, Tr=np.hstack([Rv, tv[:, None]]), ref_pts=ref_pts.numpy(),
                        uvd=ref_uvd, C2V=rc.C2V.numpy())
    os.remove(calib_path)

    # ---------------- G7: end-to-end tiny scene -> final detection JSON (oracle pipeline over on-disk inputs)
    import tempfile as _tf
    from cm3d_amd import nusc_io
    from tests.helpers import oracle_results
    with _tf.TemporaryDirectory() as td:
        tiny = syn.config("tiny")
        dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(td, tiny, n_scenes=1, frames_per_scene=2)
        tables = nusc_io.NuscTables("v1.0-synth", dataroot)
        scene = tables.scene_by_name(names[0])
        frames = nusc_io.frames_of_scene(tables, scene, mask_dir, n_sweeps=3, ratio=tiny.ratio)
        lanes7 = [nusc_io.load_lane_points(dataroot, tables.location(scene))]
        res7 = oracle_results(orc, frames, lanes7, [0] * len(frames))
    g7 = {"meta": {"use_camera": True, "use_lidar": False, "use_radar": False, "use_map": True, "use_external": False}, "results": res7}
    json.dump(g7, open(os.path.join(HERE, "g7_tiny_scene.json"), "w"))
    report["G7 boxes in the tiny scene"] = int(sum(len(v) for v in res7.values()))

    # ---------------- small helpers of the reference
    report["get_detection_name"] = {k: ref.get_detection_name(k) for k in ["trafficcone", "constructionvehicle", "human", "car"]}
    json.dump(report, open(os.path.join(HERE, "gen_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
