"""Seeded synthetic nuScenes-/Waymo-shaped inputs for the lifting path.

Shapes follow BASELINE.md section 3 / SURVEY.md section 8(d): a LiDAR-like ring pattern in
the sensor frame, a rigid ego pose with translation ~(600,1600,0) m so that the
global-frame float32 cancellation of the reference is exercised, a 6-camera rig,
n instance masks per frame given as COCO RLE (the on-disk format, reference
src/nuscenes/gen_2d_masks_detic.py:468-506), class labels from the 10-class prior
table and scores rounded to two decimals.
Everything is numpy on the host and deterministic in (seed, frame index).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import geometry as geo
from . import rle as rlemod

CLASSES = ["car", "truck", "bus", "trailer", "construction_vehicle", "pedestrian",
           "motorcycle", "bicycle", "traffic_cone", "barrier"]
# labels as the mask producer spells them (reference 2d_to_3d.py:122-132 renames three)
PRODUCER_LABELS = ["car", "truck", "bus", "trailer", "constructionvehicle", "human",
                   "motorcycle", "bicycle", "trafficcone", "barrier"]

NUSC_CAM_YAWS_DEG = [0.0, -55.0, -110.0, 180.0, 110.0, 55.0]   # CAM_LIST order, 2d_to_3d.py:62-69
_CAM_BASE = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)  # cam (x right,y down,z fwd) -> ego


@dataclass
class SyntheticConfig:
    n_points: int = 35000          # points per sweep
    n_sweeps: int = 1
    n_masks: int = 20
    n_cams: int = 6
    width: int = 1600              # mask width  (W_img * ratio)
    height: int = 900
    ratio: float = 1.0             # intrinsics scale (0.64 in the reference, 2d_to_3d.py:419)
    n_beams: int = 32
    focal: float = 1266.4
    full_width: int = 1600
    full_height: int = 900
    min_area: float = 200.0
    max_area: float = 60000.0
    seed: int = 1234
    empty_mask_prob: float = 0.1   # masks dropped on sky/empty regions
    duplicate_prob: float = 0.2    # masks that re-detect an object of an earlier mask
    point_order: str = "ring"      # row order of a sweep: "ring" = beam by beam, each a full turn (KITTI / Waymo range
                                   # image rows); "firing" = azimuth step by azimuth step, all beams of one firing
                                   # together (how the HDL-32E packets of a nuScenes LIDAR_TOP .pcd.bin are laid out,
                                   # the ring index in column 4 cycling 0..31)
    ego_magnitude: float = None    # distance of the ego pose from the map origin in metres (None: the ~1.7 km of
                                   # _ego_pose's default; 0: a vehicle-frame dataset like Waymo/KITTI; nuScenes maps
                                   # reach ~4 km).  What the float32 global-frame cancellation scales with.


@dataclass
class Frame:
    token: str
    sweeps_raw: List[np.ndarray]      # each (n,5) float32, sensor frame
    sweep_xf: np.ndarray              # (n_sweeps, SWEEP_XF_STRIDE) float32
    cams: np.ndarray                  # (C, CAM_STRIDE) float32
    rles: List[dict]                  # COCO RLE dicts, size [W,H]
    labels: List[str]
    scores: List[float]
    cam_nums: List[int]
    ego_xyz: np.ndarray               # (3,) float64, LIDAR_TOP ego_pose translation of the keyframe
    width: int
    height: int
    meta: dict = field(default_factory=dict)


def _ego_pose(rng, magnitude=None):
    if magnitude is None:
        cx, cy, jit = 600.0, 1600.0, 200.0
    else:       # same direction from the origin as the default, `magnitude` metres out; the same three draws
        cx, cy = 0.3511234 * float(magnitude), 0.9363291 * float(magnitude)
        jit = 200.0 if magnitude >= 1000.0 else 5.0
    t = np.array([cx + rng.uniform(-jit, jit), cy + rng.uniform(-jit, jit), rng.uniform(0.0, 2.0)])
    yaw = rng.uniform(-np.pi, np.pi)
    R = geo.rot_z(yaw) @ geo.quat_to_rotmat([1.0, rng.normal() * 0.004, rng.normal() * 0.004, 0.0])
    return t, geo.rotmat_to_quat(R), yaw


def _lidar_sweep(rng, cfg: SyntheticConfig, objects):
    """Sensor-frame ring pattern: ground returns for downward beams, sparse
    structure returns above, closer returns on the synthetic objects."""
    n = cfg.n_points
    nb = cfg.n_beams
    per = n // nb
    elev = np.deg2rad(np.linspace(-30.67, 10.67, nb))
    az = np.tile(np.linspace(-np.pi, np.pi, per, endpoint=False), nb)[: nb * per]
    az = az + rng.normal(scale=2e-4, size=az.size)
    el = np.repeat(elev, per)
    h = 1.84
    with np.errstate(divide="ignore"):
        r_ground = np.where(el < -0.005, h / np.sin(-el), np.inf)
    r_far = rng.uniform(8.0, 70.0, size=az.size)
    r = np.minimum(r_ground, r_far)
    r = np.minimum(r, 100.0)
    # objects: vertical cylinders in the sensor frame
    for (oa, od, orad, oh) in objects:
        da = np.angle(np.exp(1j * (az - oa)))
        half = np.arctan2(orad, od)
        zhit = h + od * np.tan(el)                    # height above ground at distance od
        hit = (np.abs(da) < half) & (od / np.cos(el) < r) & (zhit > 0.0) & (zhit < oh)
        r = np.where(hit, od / np.cos(el) + rng.normal(scale=0.03, size=az.size), r)
    r = r * (1.0 + rng.normal(scale=1e-3, size=az.size))
    x = r * np.cos(el) * np.cos(az)
    y = r * np.cos(el) * np.sin(az)
    z = r * np.sin(el)
    pts = np.stack([x, y, z, rng.uniform(0, 255, size=az.size), np.repeat(np.arange(nb), per).astype(np.float64)], 1)
    if cfg.point_order == "firing":
        pts = np.ascontiguousarray(pts.reshape(nb, per, 5).transpose(1, 0, 2)).reshape(nb * per, 5)
    elif cfg.point_order != "ring":
        raise ValueError(f"unknown point_order {cfg.point_order!r}")
    extra = n - pts.shape[0]
    if extra > 0:      # a few returns inside the ego box so that the a2 filter has work to do
        e = np.stack([rng.uniform(-1.4, 1.4, extra), rng.uniform(-1.4, 1.4, extra), rng.uniform(-1.5, 0.5, extra),
                      rng.uniform(0, 255, extra), np.zeros(extra)], 1)
        pts = np.concatenate([pts, e], 0)
    # a handful of ego-box points in every sweep
    k = min(64, pts.shape[0])
    sel = rng.choice(pts.shape[0], k, replace=False)
    pts[sel, 0] = rng.uniform(-1.5, 1.5, k)
    pts[sel, 1] = rng.uniform(-1.5, 1.5, k)
    return pts.astype(np.float32)


def _ellipse_rle(cx, cy, ax, ay, W, H):
    y0 = max(int(np.ceil(cy - ay)), 0)
    y1 = min(int(np.floor(cy + ay)), H - 1)
    if y1 < y0:
        return rlemod.spans_to_counts([], [], [], W, H)
    ys = np.arange(y0, y1 + 1)
    hw = ax * np.sqrt(np.maximum(0.0, 1.0 - ((ys - cy) / ay) ** 2))
    x0 = np.maximum(np.ceil(cx - hw).astype(np.int64), 0)
    x1 = np.minimum(np.floor(cx + hw).astype(np.int64), W - 1)
    ok = x1 >= x0
    return rlemod.spans_to_counts(ys[ok], x0[ok], x1[ok], W, H)


def make_frame(cfg: SyntheticConfig, index: int) -> Frame:
    rng = np.random.default_rng(cfg.seed + index)
    W, H = cfg.width, cfg.height
    # --- rig
    lidar_cs_t = np.array([0.943713, 0.0, 1.84023])
    lidar_cs_q = geo.rotmat_to_quat(geo.rot_z(np.deg2rad(-89.85)) @ geo.quat_to_rotmat([1.0, 0.003, -0.002, 0.0]))
    ego_t, ego_q, ego_yaw = _ego_pose(rng, cfg.ego_magnitude)
    # --- objects (sensor-frame cylinders), one per mask that is not "empty"
    n = cfg.n_masks
    cls = rng.integers(0, len(CLASSES), size=n)
    objects = []
    obj_of_mask = []
    for i in range(n):
        u = rng.uniform()
        if u < cfg.empty_mask_prob:
            obj_of_mask.append(None)
            continue
        if u > 1.0 - cfg.duplicate_prob and objects:
            # a second detection of an object that already has a mask (same class): circle-NMS work
            j = int(rng.integers(0, i))
            if obj_of_mask[j] is not None:
                obj_of_mask.append(obj_of_mask[j])
                cls[i] = cls[j]
                continue
        oa = rng.uniform(-np.pi, np.pi)
        od = np.exp(rng.uniform(np.log(4.0), np.log(55.0)))
        orad = rng.uniform(0.2, 1.6)
        oh = rng.uniform(0.8, 3.0)
        objects.append((oa, od, orad, oh))
        obj_of_mask.append(len(objects) - 1)
    # --- sweeps (each with its own ego pose, the vehicle creeps forward)
    sweeps_raw, sweep_xf = [], []
    for s in range(cfg.n_sweeps):
        dt = 0.05 * s
        fwd = geo.quat_to_rotmat(ego_q) @ np.array([8.0 * dt, 0.0, 0.0])
        sweeps_raw.append(_lidar_sweep(rng, cfg, objects))
        sweep_xf.append(geo.sweep_xf_record(lidar_cs_t, lidar_cs_q, ego_t + fwd, ego_q))
    # --- cameras (ego pose at the camera timestamp differs slightly from the lidar's)
    cams = []
    K = np.array([[cfg.focal, 0.0, cfg.full_width * 0.51], [0.0, cfg.focal, cfg.full_height * 0.546], [0.0, 0.0, 1.0]])
    cam_R, cam_t, cam_ego_t = [], [], []
    for c in range(cfg.n_cams):
        yaw = np.deg2rad(NUSC_CAM_YAWS_DEG[c % 6] + (0.0 if c < 6 else 30.0))
        Rcs = geo.rot_z(yaw) @ _CAM_BASE @ geo.quat_to_rotmat([1.0, rng.normal() * 0.005, rng.normal() * 0.005, rng.normal() * 0.005])
        tcs = geo.rot_z(yaw) @ np.array([1.5, 0.0, 0.0]) + np.array([0.2, 0.0, 1.51])
        ego_c = ego_t + geo.quat_to_rotmat(ego_q) @ np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.02, 0.02), 0.0])
        cams.append(geo.nusc_cam_record(ego_c, ego_q, tcs, geo.rotmat_to_quat(Rcs), K, cfg.ratio))
        cam_R.append(Rcs); cam_t.append(tcs); cam_ego_t.append(ego_c)
    cams = np.stack(cams)
    # --- masks: ellipse around the projection of each object (or a random one)
    Rl = geo.quat_to_rotmat(lidar_cs_q)
    Re = geo.quat_to_rotmat(ego_q)
    f = cfg.focal * cfg.ratio
    rles, labels, scores, cam_nums = [], [], [], []
    for i in range(n):
        area = np.exp(rng.uniform(np.log(cfg.min_area), np.log(cfg.max_area)))
        aspect = np.exp(rng.uniform(np.log(0.4), np.log(2.5)))
        placed = False
        if obj_of_mask[i] is not None:
            oa, od, orad, oh = objects[obj_of_mask[i]]
            p_s = np.array([od * np.cos(oa), od * np.sin(oa), -1.84 + 0.5 * oh])
            p_e = Rl @ p_s + lidar_cs_t
            best = None
            for c in range(cfg.n_cams):
                pc = cam_R[c].T @ (p_e - cam_t[c])
                if pc[2] > 1.0:
                    u = f * pc[0] / pc[2] + K[0, 2] * cfg.ratio
                    v = f * pc[1] / pc[2] + K[1, 2] * cfg.ratio
                    if 0 <= u < W and 0 <= v < H and (best is None or pc[2] < best[3]):
                        best = (c, u, v, pc[2])
            if best is not None:
                c, u, v, d = best
                ax = max(2.0, f * orad / d * rng.uniform(0.8, 1.4))
                ay = max(2.0, f * 0.5 * oh / d * rng.uniform(0.8, 1.3))
                # honour the area range of the config
                sc = np.sqrt(np.clip(np.pi * ax * ay, cfg.min_area, cfg.max_area) / (np.pi * ax * ay))
                ax, ay = ax * sc, ay * sc
                cx, cy = u + rng.normal() * 0.15 * ax, v + rng.normal() * 0.15 * ay
                placed = True
        if not placed:
            c = int(rng.integers(0, cfg.n_cams))
            ax = np.sqrt(area * aspect / np.pi)
            ay = area / (np.pi * ax)
            cx, cy = rng.uniform(0, W), rng.uniform(0, H * (0.35 if obj_of_mask[i] is None else 1.0))
        cnts = _ellipse_rle(cx, cy, ax, ay, W, H)
        rles.append({"size": [W, H], "counts": rlemod.counts_to_string(cnts)})
        labels.append(PRODUCER_LABELS[int(cls[i])])
        scores.append(float(np.round(rng.uniform(0.3, 1.0), 2)))
        cam_nums.append(int(c))
    # a mask touching the image border and one degenerate mask keep the edge paths alive
    if n >= 4:
        rles[n - 1] = {"size": [W, H], "counts": rlemod.counts_to_string(_ellipse_rle(3.0, H * 0.6, 40.0, 60.0, W, H))}
        rles[n - 2] = {"size": [W, H], "counts": rlemod.counts_to_string(_ellipse_rle(W * 0.5, H * 0.55, 1.0, 1.0, W, H))}
    return Frame(token=f"synthetic-{cfg.seed}-{index:06d}", sweeps_raw=sweeps_raw, sweep_xf=np.stack(sweep_xf),
                 cams=cams, rles=rles, labels=labels, scores=scores, cam_nums=cam_nums,
                 ego_xyz=ego_t.copy(), width=W, height=H,
                 meta={"ego_yaw": ego_yaw, "objects": _ground_truth_objects(objects, obj_of_mask, cls, sweep_xf[0])})


def _ground_truth_objects(objects, obj_of_mask, cls, xf0):
    """The synthetic cylinders as ground-truth boxes in the global frame of the key sweep: what an annotator would have
    drawn (centre, size w = l = diameter, h, class of the first mask that shows the object).  Used by the synthetic
    dataset writer to emit sample_annotation rows for the evaluation harness; draws no random numbers."""
    xf0 = np.asarray(xf0, np.float64)
    R_cs, t_cs, R_ego, t_ego = xf0[0:9].reshape(3, 3), xf0[9:12], xf0[12:21].reshape(3, 3), xf0[21:24]
    out = []
    for j, (oa, od, orad, oh) in enumerate(objects):
        first = next(i for i, o in enumerate(obj_of_mask) if o == j)
        p_s = np.array([od * np.cos(oa), od * np.sin(oa), -1.84 + 0.5 * oh])
        p_g = R_ego @ (R_cs @ p_s + t_cs) + t_ego
        out.append({"center": p_g.tolist(), "size": [2.0 * orad, 2.0 * orad, float(oh)], "label": PRODUCER_LABELS[int(cls[first])]})
    return out


def make_waymo_frame(cfg: SyntheticConfig, index: int):
    """A Waymo-shaped frame (reference src/waymo/2d_to_3d.py): one vehicle-frame cloud, cameras given as
    4x4 extrinsics in Waymo's camera axes (x forward, y left, z up) + [fx, fy, cx, cy] intrinsics, a frame
    pose (vehicle -> global).  Built on make_frame so masks and points stay consistent."""
    from . import waymo as wm
    fr = make_frame(cfg, index)
    rng = np.random.default_rng(cfg.seed + 7919 * (index + 1))
    xf = fr.sweep_xf[0].astype(np.float64)
    R_l, t_l = xf[0:9].reshape(3, 3), xf[9:12]
    pts_vehicle = (fr.sweeps_raw[0][:, :3].astype(np.float64) @ R_l.T + t_l).astype(np.float32)
    S = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)      # :561-565
    cams = []
    for c in range(fr.cams.shape[0]):
        t_cs_neg, R_csT, _ = geo.cam_stage(fr.cams[c], 1)
        T = np.eye(4)
        T[:3, :3] = R_csT.T                             # optical camera -> vehicle
        T[:3, 3] = -t_cs_neg
        K = geo.cam_K(fr.cams[c]) / cfg.ratio
        cams.append(((T @ S).reshape(16), [K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, 0, 0, 0]))
    yaw = rng.uniform(-np.pi, np.pi)
    P = np.eye(4)
    P[:3, :3] = geo.rot_z(yaw) @ geo.quat_to_rotmat([1.0, rng.normal() * 0.01, rng.normal() * 0.01, 0.0])
    P[:3, 3] = [5000.0 + rng.uniform(-300, 300), -3000.0 + rng.uniform(-300, 300), 20.0 + rng.uniform(-5, 5)]
    ok = {"barrier": "car", "trafficcone": "human"}
    labels = [ok.get(l, l) for l in fr.labels]
    wf = wm.frame_from_extracted(f"waymo-{cfg.seed}-{index:06d}", pts_vehicle, cams, fr.rles, labels, fr.scores, fr.cam_nums,
                                 P.reshape(16), fr.width, fr.height, timestamp_micros=1_550_000_000_000_000 + index * 100_000,
                                 context_name=f"synthetic-context-{cfg.seed}")
    # cam records must use this config's ratio (the reference hard-codes 1024/1920 for the real dataset)
    wf.cams = np.stack([wm.cam_record(e, i, ratio=cfg.ratio) for e, i in cams])
    return wf


def make_kitti_frame(cfg: SyntheticConfig, index: int):
    """A KITTI-shaped frame (reference src/kitti/2d_to_3d.py): one velodyne scan (n,4), one camera given by
    P2 / R0_rect / Tr_velo_to_cam.  Returns (frame, calibs) where calibs holds the three float64 matrices."""
    import torch
    from . import kitti as kt
    one = SyntheticConfig(**{**cfg.__dict__, "n_cams": 1, "n_sweeps": 1})
    fr = make_frame(one, index)
    xf = fr.sweep_xf[0].astype(np.float64)
    velo = np.concatenate([(fr.sweeps_raw[0][:, :3].astype(np.float64) @ xf[0:9].reshape(3, 3).T + xf[9:12]),
                           fr.sweeps_raw[0][:, 3:4] / 255.0], 1).astype(np.float32)
    t_cs_neg, R_csT, _ = geo.cam_stage(fr.cams[0], 1)
    Tr = np.hstack([R_csT, (R_csT @ t_cs_neg)[:, None]])          # velo -> camera: R^T (p - t)
    R0 = geo.rot_z(0.0015) @ geo.quat_to_rotmat([1.0, 0.002, -0.0015, 0.0])
    K = geo.cam_K(fr.cams[0]) / cfg.ratio
    P2 = np.hstack([K, [[44.857], [0.2164], [0.002746]]])
    P2[2, 2] = 1.0
    calibs = {"P2": torch.tensor(P2.reshape(-1), dtype=torch.float32), "R0_rect": torch.tensor(R0.reshape(-1), dtype=torch.float32),
              "Tr_velo_to_cam": torch.tensor(Tr.reshape(-1), dtype=torch.float32)}
    kf = kt.frame_from_arrays(index, velo, kt.Calibration(calibs=calibs), fr.rles, fr.labels, fr.scores, ratio=cfg.ratio)
    return kf, {"P2": P2, "R0_rect": R0, "Tr_velo_to_cam": Tr}


def make_lane_table(center_xy, n_points=50000, seed=0, extent=200.0):
    """Synthetic HD-map lane centre-lines discretised at 0.5 m as (x, y, yaw) rows
    (what nuscenes-devkit's discretize_lanes yields, reference 2d_to_3d.py:228-240)."""
    rng = np.random.default_rng(seed)
    per_lane = int(2 * extent / 0.5)
    n_lanes = max(1, n_points // per_lane)
    rows = []
    for k in range(n_lanes):
        yaw = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2]) + rng.normal() * 0.05
        off = rng.uniform(-extent, extent)
        s = np.arange(per_lane) * 0.5 - extent
        curve = 0.0005 * rng.normal() * s * s
        x = s * np.cos(yaw) - (off + curve) * np.sin(yaw)
        y = s * np.sin(yaw) + (off + curve) * np.cos(yaw)
        rows.append(np.stack([x + center_xy[0], y + center_xy[1], np.full_like(s, np.angle(np.exp(1j * yaw)))], 1))
    lane = np.concatenate(rows, 0)
    if lane.shape[0] < n_points:
        pad = lane[rng.integers(0, lane.shape[0], n_points - lane.shape[0])] + rng.normal(scale=0.2, size=(n_points - lane.shape[0], 3)) * [1, 1, 0]
        lane = np.concatenate([lane, pad], 0)
    return lane[:n_points].astype(np.float64)


# named configurations of BASELINE.md section 3
def config(name: str, **over) -> SyntheticConfig:
    base = {
        "c1": dict(n_points=34700, n_sweeps=3, n_masks=20, n_cams=6, width=1024, height=576, ratio=0.64, point_order="firing"),
        "c2": dict(n_points=35000, n_sweeps=1, n_masks=20, n_cams=6, width=1600, height=900, ratio=1.0, point_order="firing"),
        "c4": dict(n_points=180000, n_sweeps=1, n_masks=20, n_cams=5, width=1920, height=1280, ratio=1.0,
                   n_beams=64, focal=2060.0, full_width=1920, full_height=1280),
        "c5": dict(n_points=35000, n_sweeps=10, n_masks=80, n_cams=6, width=1600, height=900, ratio=1.0, point_order="firing"),
        "tiny": dict(n_points=3000, n_sweeps=2, n_masks=8, n_cams=6, width=256, height=144, ratio=0.16,
                     min_area=30.0, max_area=3000.0),
    }[name]
    base.update(over)
    return SyntheticConfig(**base)
