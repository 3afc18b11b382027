"""KITTI-side host pieces of the lifting path (reference src/kitti/2d_to_3d.py,
src/kitti/kitti_utils.py).

The committed KITTI script stops at a debug `print(...); exit()` (2d_to_3d.py:1528) and its second
stage references undefined names; what is built here is its evident intent: stage 1 writing one KITTI
label line per mask with at least 4 in-mask points -- medoid centre (ref-camera frame), class prior
dimensions `(h,w,l) = prior[[2,0,1]]`, `y += h/2`, yaw of a PCA box of the in-mask points
(:1479-1536, :855-885).

Differences against nuScenes that matter to the kernels:
  * the cloud lives in the reference-camera frame: velo -> ref with Tr_velo_to_cam (:1066-1073);
  * the per-mask chain is ref -> velo -> ref -> rect (:1238-1240, kitti_utils.py:224-249), each step a
    homogeneous (n x 4) @ (4 x 3) product = rotate then translate -> a 3-stage camera record;
  * one camera, no cam_nums, ratio 0.8366 (:996), masks with <= 3 points are skipped (:1479-1480).
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import geometry as geo

RATIO = 0.8366          # :996, :1103


class Calibration:
    """kitti_utils.Calibration (:147-170) restated on float32 tensors: P2, Tr_velo_to_cam (V2C), its rigid
    inverse C2V (inverse_rigid_trans :368-375) and R0_rect."""

    def __init__(self, calib_filepath=None, calibs=None):
        calibs = calibs if calibs is not None else self.read_calib_file(calib_filepath)
        self.P = calibs["P2"].view([3, 4])
        self.V2C = calibs["Tr_velo_to_cam"].view([3, 4])
        self.C2V = inverse_rigid_trans(self.V2C)
        self.R0 = calibs["R0_rect"].view([3, 3])
        self.c_u, self.c_v = self.P[0, 2], self.P[1, 2]
        self.f_u, self.f_v = self.P[0, 0], self.P[1, 1]

    @staticmethod
    def read_calib_file(filepath):
        data = {}
        with open(filepath, "r") as f:
            for line in f.readlines():
                line = line.rstrip()
                if len(line) == 0:
                    continue
                key, value = line.split(":", 1)
                try:
                    data[key] = torch.from_numpy(np.array([float(x) for x in value.split()])).to(dtype=torch.float32)
                except ValueError:
                    pass
        return data

    def cam_record(self, ratio=RATIO):
        """3-stage record: ref -> velo (C2V), velo -> ref (V2C), ref -> rect (R0); K' = f32([[f_u,0,c_u],[0,f_v,c_v],[0,0,1]]) * ratio
        with K'[2][2] = 1 (:1246-1256)."""
        K = torch.tensor([[float(self.f_u), 0, float(self.c_u)], [0, float(self.f_v), float(self.c_v)], [0, 0, 1]], dtype=torch.float32)
        K = K * ratio
        K[2, 2] = 1
        c2v, v2c = self.C2V.numpy(), self.V2C.numpy()
        return geo.make_cam_record([(None, c2v[:, :3], c2v[:, 3]), (None, v2c[:, :3], v2c[:, 3]), (None, self.R0.numpy(), None)], K.numpy())

    def sweep_xf(self):
        """velo -> ref as a sweep transform record: rotate with V2C_R, translate by V2C_t, then identity."""
        r = np.zeros(geo.SWEEP_XF_STRIDE, np.float32)
        v2c = self.V2C.numpy()
        r[0:9] = v2c[:, :3].reshape(9)
        r[9:12] = v2c[:, 3]
        r[12:21] = np.eye(3, dtype=np.float32).reshape(9)
        return r


def inverse_rigid_trans(Tr):
    """kitti_utils.py:368-375 on float32 tensors."""
    inv = torch.zeros_like(Tr)
    inv[0:3, 0:3] = Tr[0:3, 0:3].transpose(0, 1)
    inv[0:3, 3] = torch.matmul(-Tr[0:3, 0:3].transpose(0, 1), Tr[0:3, 3])
    return inv


def frame_from_files(frame_num, velo_path, calib_path, rles, labels, scores, ratio=RATIO):
    """Kernel inputs of one KITTI frame (:1001-1008, :1066-1073)."""
    calib = Calibration(calib_path)
    velo = np.fromfile(velo_path, dtype=np.float32).reshape(-1, 4)
    return frame_from_arrays(frame_num, velo, calib, rles, labels, scores, ratio)


def frame_from_arrays(frame_num, velo, calib, rles, labels, scores, ratio=RATIO):
    W, H = rles[0]["size"] if rles else (int(1224 * ratio), int(370 * ratio))
    return SimpleNamespace(token=f"{int(frame_num):06d}", sweeps_raw=[np.ascontiguousarray(velo, np.float32)],
                           sweep_xf=calib.sweep_xf()[None], cams=calib.cam_record(ratio)[None], rles=list(rles), labels=list(labels),
                           scores=list(scores), cam_nums=[0] * len(rles), ego_xyz=np.zeros(3), width=int(W), height=int(H),
                           pose=None, no_ego_box=True)


def obb_yaw(pts3d):
    """Yaw of a PCA box of the in-mask points: what :855-876 + :1524 extract from Open3D's
    get_oriented_bounding_box (Open3D 0.15 is not in the reference checkout; its OBB is a PCA of the convex-hull
    vertices, restated here; the sign conventions of its eigen-solver are not reproduced -- best effort, parity unpinned).
    Axis re-ordering by extent and as_euler('zyx')[0] follow the reference."""
    from scipy.spatial import ConvexHull
    from scipy.spatial.transform import Rotation
    p = np.asarray(pts3d, np.float64)
    # Open3D (0.13 - 0.15) runs the PCA on the vertices of the convex hull (Qhull, like scipy's), with the population
    # covariance; a degenerate cloud makes Qhull fail there as here, and the caller falls back to the identity box (:1481-1484)
    hull = p[ConvexHull(p).vertices]
    mean = hull.mean(0)
    cov = (hull - mean).T @ (hull - mean) / hull.shape[0]
    w, v = np.linalg.eigh(cov)
    Rm = v[:, ::-1].copy()                       # columns: largest variance first
    if np.linalg.det(Rm) < 0:
        Rm[:, 2] = -Rm[:, 2]
    size = p.max(0) - p.min(0)
    axis = [a for _, a in sorted(zip(size, "xyz"), key=lambda t: t[0])]
    Rm = np.stack([Rm[:, axis.index("z")], Rm[:, axis.index("y")], Rm[:, axis.index("x")]], axis=1)
    if np.linalg.det(Rm) < 0:
        Rm[:, 0] = -Rm[:, 0]
    return float(Rotation.from_matrix(Rm).as_euler("zyx")[0])


# src/kitti/2d_to_3d.py:105-116: what get_detection_name (:183-197) finally returns -- the KITTI class written to the label file
KITTI_CLASS_MAPS = {"car": "Car", "pedestrian": "Pedestrian", "truck": "Truck", "bus": "Tram", "traffic_cone": "Misc",
                    "construction_vehicle": "Misc", "bicycle": "Cyclist", "motorcycle": "Cyclist", "trailer": "Misc", "barrier": "Misc"}


def label_line(object_type, wlh, xyz, yaw, conf=None, truncation=-1, occlusion=-1, alpha=-10):
    """save_pred (:879-885); the 2D box is written as 0 0 0 0 (:1535-1536)."""
    ltrb = [0, 0, 0, 0]
    s = (f"{object_type} {truncation} {occlusion} {alpha} {ltrb[0]} {ltrb[1]} {ltrb[2]} {ltrb[3]} "
         f"{wlh[0]} {wlh[1]} {wlh[2]} {xyz[0]} {xyz[1]} {xyz[2]} {yaw}")
    return s + (f" {conf}\n" if conf is not None else "\n")


def labels_of_frame(hb, res, f, classes, shape_priors):
    """Label lines (pred with score, pseudo without) of frame f from the device results (:1479-1536)."""
    from .lifting import get_detection_name
    pred, pseudo = [], []
    for m in range(hb.mask_off[f], hb.mask_off[f + 1]):
        o, e = res["hit_off"][m], res["hit_off"][m + 1]
        if e - o <= 3:                                   # :1479-1480
            continue
        pts = res["hit_xyz"][o:e, :3]                    # the in-mask points (:1479), laid out like hit_idx
        try:
            yaw = obb_yaw(pts)
        except Exception:                                # :1481-1484: bare except -> identity box
            yaw = 0.0
        label = hb.labels[f][m - hb.mask_off[f]]
        name = KITTI_CLASS_MAPS[get_detection_name(label)]       # :1523, :183-197: the label file carries the KITTI class
        # :1530 looks the prior up by the RAW label; for the three spellings get_detection_name renames (trafficcone,
        # constructionvehicle, human) the reference's table has no key and it raises -- here the renamed key is used
        wlh = shape_priors[label] if label in shape_priors else shape_priors[get_detection_name(label)]
        wlh = [wlh[2], wlh[0], wlh[1]]                   # :1530-1531
        c = [float(v) for v in res["centroid"][m]]
        center = [c[0], c[1] + wlh[0] / 2, c[2]]         # :1533
        score = hb.score[m]
        pred.append(label_line(name, wlh, center, yaw, score))
        pseudo.append(label_line(name, wlh, center, yaw, None))
    return pred, pseudo
