"""ctypes front-end of the CPU oracle (oracle/cm3d_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under cm3d_amd/ may import this module.

`lift_frame_reference_order` runs one frame the way the reference's per-mask
loop does (src/nuscenes/2d_to_3d.py:510-667 of the reference checkout).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcm3d_oracle.so")

CAM_STRIDE = 64
MIN_DIST_F32 = np.float32(2.3)                 # 2d_to_3d.py:348,598
EGO_HALFW_F32 = np.float32(np.sqrt(2.3))       # 2d_to_3d.py:443-444


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "cm3d_oracle.c")):
        build()
    lib = C.CDLL(_SO)
    p = C.c_void_p
    i64, i32, f32 = C.c_int64, C.c_int, C.c_float
    lib.orc_sweep_prep.restype = i64
    lib.orc_sweep_prep.argtypes = [p, i64, i32, p, p, p, p, f32, p]
    lib.orc_rle_string_to_counts.restype = i64
    lib.orc_rle_string_to_counts.argtypes = [C.c_char_p, i64, p, i64]
    lib.orc_rle_counts_to_string.restype = i64
    lib.orc_rle_counts_to_string.argtypes = [p, i64, p, i64]
    lib.orc_rle_to_dense.restype = i32
    lib.orc_rle_to_dense.argtypes = [p, i64, i64, p]
    lib.orc_dense_to_rle.restype = i64
    lib.orc_dense_to_rle.argtypes = [p, i64, p, i64]
    lib.orc_erode3x3.restype = None
    lib.orc_erode3x3.argtypes = [p, i32, i32, p]
    lib.orc_points_in_mask.restype = i64
    lib.orc_points_in_mask.argtypes = [p, i64, p, p, i32, i32, f32, p, p]
    lib.orc_project_points.restype = None
    lib.orc_project_points.argtypes = [p, i64, p, p]
    lib.orc_medoid.restype = i64
    lib.orc_medoid.argtypes = [p, p, i64, p]
    lib.orc_lane_nn.restype = None
    lib.orc_lane_nn.argtypes = [p, i64, p, i64, p, p]
    lib.orc_box_assemble.restype = None
    lib.orc_box_assemble.argtypes = [p, p, f32, p, i32, p, p]
    lib.orc_centroid_transform.restype = None
    lib.orc_centroid_transform.argtypes = [p, p, p]
    lib.orc_box_assemble_waymo.restype = None
    lib.orc_box_assemble_waymo.argtypes = [p, p, p, f32, i32, p, p]
    lib.orc_bev_iou.restype = C.c_double
    lib.orc_bev_iou.argtypes = [p, p]
    lib.orc_bev_match.restype = i64
    lib.orc_bev_match.argtypes = [p, i32, p, i32, C.c_double, p, p, p, p]
    lib.orc_circle_nms.restype = i32
    lib.orc_circle_nms.argtypes = [p, p, p, p, i32, p, p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- a2
def sweep_prep(raw, R_cs, t_cs, R_ego, t_ego, halfw=EGO_HALFW_F32):
    """raw (n,5|4) f32 -> (m,4) f32 global-frame points, ego box removed."""
    raw = _f32(raw)
    out = np.empty((raw.shape[0], 4), np.float32)
    n = lib().orc_sweep_prep(_ptr(raw), raw.shape[0], raw.shape[1], _ptr(_f32(R_cs)), _ptr(_f32(t_cs)),
                             _ptr(_f32(R_ego)), _ptr(_f32(t_ego)), float(halfw), _ptr(out))
    return out[:n].copy()


# ---------------------------------------------------------------- a1
def rle_string_to_counts(s: bytes):
    cnts = np.empty(len(s) + 1, np.uint32)
    m = lib().orc_rle_string_to_counts(s, len(s), _ptr(cnts), cnts.size)
    if m < 0:
        raise ValueError("malformed RLE string")
    return cnts[:m].copy()


def rle_counts_to_string(cnts) -> bytes:
    cnts = np.ascontiguousarray(cnts, np.uint32)
    buf = np.empty(cnts.size * 7 + 8, np.uint8)
    n = lib().orc_rle_counts_to_string(_ptr(cnts), cnts.size, _ptr(buf), buf.size)
    assert n >= 0
    return buf[:n].tobytes()


def rle_decode(rle):
    """COCO RLE dict {'size':[h,w],'counts':bytes} -> (h,w) uint8 F-ordered,
    like pycocotools.mask.decode for one object (here h=W_img, w=H_img)."""
    h, w = rle["size"]
    cnts = rle_string_to_counts(rle["counts"])
    flat = np.empty(h * w, np.uint8)
    rc = lib().orc_rle_to_dense(_ptr(cnts), cnts.size, h * w, _ptr(flat))
    if rc != 0:
        raise ValueError("RLE run lengths do not match size")
    return flat.reshape((h, w), order="F")


def rle_encode(arr_hw_fortran):
    """inverse of rle_decode: (h,w) array (any order) -> COCO RLE dict."""
    a = np.asfortranarray(arr_hw_fortran)
    h, w = a.shape
    flat = np.ascontiguousarray(a.reshape(-1, order="F"), np.uint8)
    cnts = np.empty(flat.size + 1, np.uint32)
    m = lib().orc_dense_to_rle(_ptr(flat), flat.size, _ptr(cnts), cnts.size)
    return {"size": [h, w], "counts": rle_counts_to_string(cnts[:m])}


# ---------------------------------------------------------------- a3
def erode3x3(img_hw):
    img = np.ascontiguousarray(img_hw, np.uint8)
    out = np.empty_like(img)
    lib().orc_erode3x3(_ptr(img), img.shape[0], img.shape[1], _ptr(out))
    return out


# ---------------------------------------------------------------- a4-a8
def make_cam(t1, R1, t2, R2, K, stages=2):
    """Assemble one camera record (float32[CAM_STRIDE]); all inputs are already
    the float32 tensors the reference feeds to translate/rotate/view_points."""
    c = np.zeros(CAM_STRIDE, np.float32)
    c[0:3] = _f32(t1).reshape(3)
    c[3:12] = _f32(R1).reshape(9)
    c[15:18] = _f32(t2).reshape(3)
    c[18:27] = _f32(R2).reshape(9)
    c[45:54] = _f32(K).reshape(9)
    c[54] = stages
    c[55] = 1 | (4 if stages > 1 else 0)       # both stages translate before they rotate
    return c


def points_in_mask(pts, cam, eroded_hw, min_dist=MIN_DIST_F32):
    pts = _f32(pts)
    N = pts.shape[0]
    er = np.ascontiguousarray(eroded_hw, np.uint8)
    H, W = er.shape
    out = np.empty(max(N, 1), np.int32)
    scratch = np.empty(3 * max(N, 1), np.float32)
    m = lib().orc_points_in_mask(_ptr(pts), N, _ptr(_f32(cam)), _ptr(er), W, H, float(min_dist), _ptr(out), _ptr(scratch))
    return out[:m].copy()


def project_points(pts, cam):
    """(N,3) float32 rows (u, v, camera-frame depth)."""
    pts = _f32(pts)
    out = np.empty((pts.shape[0], 3), np.float32)
    lib().orc_project_points(_ptr(pts), pts.shape[0], _ptr(_f32(cam)), _ptr(out))
    return out


# ---------------------------------------------------------------- a9
def medoid(pts, idx, want_colsum=False):
    pts = _f32(pts)
    idx = np.ascontiguousarray(idx, np.int32)
    cs = np.empty(max(idx.size, 1), np.float32)
    j = lib().orc_medoid(_ptr(pts), _ptr(idx), idx.size, _ptr(cs))
    return (int(j), cs[:idx.size]) if want_colsum else int(j)


# ---------------------------------------------------------------- a10
def lane_nn(centroids, lane_pts):
    cent = _f32(centroids).reshape(-1, 3)
    lane = _f32(lane_pts).reshape(-1, 3)
    K = cent.shape[0]
    j = np.empty(K, np.int32)
    d = np.empty(K, np.float64)
    lib().orc_lane_nn(_ptr(cent), K, _ptr(lane), lane.shape[0], _ptr(j), _ptr(d))
    return j, d


# ---------------------------------------------------------------- a13
def box_assemble(centroid, prior, yaw, ego_xyz, is_vehicle):
    c = _f32(centroid).reshape(3)
    pr = np.ascontiguousarray(prior, np.float64)
    ego = np.ascontiguousarray(ego_xyz, np.float64)
    t = np.empty(3, np.float64)
    q = np.empty(4, np.float64)
    lib().orc_box_assemble(_ptr(c), _ptr(pr), float(np.float32(yaw)), _ptr(ego), int(bool(is_vehicle)), _ptr(t), _ptr(q))
    return t, q


# ---------------------------------------------------------------- a15
def circle_nms(x, y, score, label, thr_by_label):
    x = np.ascontiguousarray(x, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    s = np.ascontiguousarray(score, np.float64)
    lab = np.ascontiguousarray(label, np.int32)
    thr = np.ascontiguousarray(thr_by_label, np.float64)
    keep = np.zeros(max(x.size, 1), np.uint8)
    lib().orc_circle_nms(_ptr(x), _ptr(y), _ptr(s), _ptr(lab), x.size, _ptr(thr), _ptr(keep))
    return keep[:x.size].astype(bool)


# ---------------------------------------------------------------- frame driver
# class tables (cfg/shape_priors_chatgpt.json:1-12, 2d_to_3d.py:70-81,763,850-861)
CLASSES = ["car", "truck", "bus", "trailer", "construction_vehicle", "pedestrian",
           "motorcycle", "bicycle", "traffic_cone", "barrier"]
PRIORS_WLH = np.array([[1.8, 4.5, 1.4], [2.6, 8.0, 3.6], [2.5, 12.0, 4.0], [2.6, 12.0, 3.6],
                       [2.0, 4.5, 2.5], [0.4, 0.7, 1.7], [0.8, 2.1, 1.7], [0.6, 1.8, 1.4],
                       [0.3, 0.3, 0.7], [0.5, 1.2, 0.9]], np.float64)
IS_VEHICLE = np.array([1, 1, 1, 1, 1, 0, 0, 0, 0, 1], bool)
NMS_THR = np.array([4, 12, 10, 10, 12, 0.175, 0.85, 0.85, 0.175, 1], np.float64)


def lift_frame_reference_order(pts, cams, masks_dense_hw, mask_cam, min_dist=MIN_DIST_F32):
    """Stage 1 of one frame in the reference's execution order: for every mask,
    erode the full frame, re-project the whole cloud, test, medoid.
    Returns (idx_lists, medoid_pos, centroids) with medoid_pos=-1 for empty masks."""
    idx_lists, med, cents = [], [], []
    for m, c in zip(masks_dense_hw, mask_cam):
        er = erode3x3(m)
        idx = points_in_mask(pts, cams[c], er, min_dist)
        idx_lists.append(idx)
        if idx.size == 0:
            med.append(-1)
            cents.append(np.full(3, np.nan, np.float32))
            continue
        j = medoid(pts, idx)
        med.append(j)
        cents.append(np.asarray(pts, np.float32)[idx[j], :3].copy())
    return idx_lists, np.array(med, np.int32), np.array(cents, np.float32).reshape(-1, 3)


def stage2_frame(centroids, med, class_id, scores, lane_pts, ego_xyz):
    """Stage 2 + NMS for one frame (2d_to_3d.py:733-822, 844-924).  Returns a dict
    of arrays over the frame's masks; `valid` marks masks that produced a box,
    `keep` the boxes surviving circle-NMS."""
    n = len(med)
    valid = np.asarray(med) >= 0
    t = np.full((n, 3), np.nan)
    q = np.zeros((n, 4))
    lane_j = np.full(n, -1, np.int32)
    lane_d = np.full(n, np.nan)
    yaw = np.zeros(n, np.float32)
    vi = np.flatnonzero(valid)
    if vi.size:
        j, d = lane_nn(centroids[vi], lane_pts)
        lane_j[vi], lane_d[vi] = j, d
        lane32 = _f32(lane_pts).reshape(-1, 3)
        yaw[vi] = lane32[j, 2]
        for k in vi:
            t[k], q[k] = box_assemble(centroids[k], PRIORS_WLH[class_id[k]], yaw[k], ego_xyz, IS_VEHICLE[class_id[k]])
    keep = np.zeros(n, bool)
    if vi.size:
        keep[vi] = circle_nms(t[vi, 0], t[vi, 1], np.asarray(scores, np.float64)[vi], np.asarray(class_id, np.int32)[vi], NMS_THR)
    return dict(valid=valid, keep=keep, translation=t, rotation=q, lane_idx=lane_j, lane_dist=lane_d, yaw=yaw)


# ---------------------------------------------------------------- a17 (Waymo)
WAYMO_NMS_GROUP = np.array([1, 1, 1, 1, 1, 2, 1, 4, 0, 0], np.int32)      # NUSC_TO_WAYMO -> Label.Type, per CLASSES entry
WAYMO_NMS_THR = np.array([1, 4, 0.175, 0.175, 0.85], np.float64)          # by Label.Type (src/waymo/2d_to_3d.py:1147-1158)


def centroid_transform(c, pose_rt):
    out = np.empty(3, np.float32)
    lib().orc_centroid_transform(_ptr(_f32(c).reshape(3)), _ptr(_f32(pose_rt)), _ptr(out))
    return out


def box_assemble_waymo(centroid_global, pose_inv, prior, yaw, is_vehicle):
    t = np.empty(3, np.float64)
    h = np.empty(1, np.float64)
    lib().orc_box_assemble_waymo(_ptr(_f32(centroid_global).reshape(3)), _ptr(_f32(pose_inv)), _ptr(np.ascontiguousarray(prior, np.float64)),
                                 float(np.float32(yaw)), int(bool(is_vehicle)), _ptr(t), _ptr(h))
    return t, float(h[0])


def stage2_frame_waymo(centroids_vehicle, med, class_id, scores, lane_pts, pose_rt, pose_inv):
    """Waymo stage 2 + NMS for one frame.  scores are the proto-float (float32) values."""
    n = len(med)
    valid = np.asarray(med) >= 0
    t = np.zeros((n, 3)); heading = np.zeros(n)
    lane_j = np.full(n, -1, np.int32); lane_d = np.full(n, np.nan); yaw = np.zeros(n, np.float32)
    cg = np.zeros((n, 3), np.float32)
    vi = np.flatnonzero(valid)
    if vi.size:
        for k in vi:
            cg[k] = centroid_transform(centroids_vehicle[k], pose_rt)
        j, d = lane_nn(cg[vi], lane_pts)
        lane_j[vi], lane_d[vi] = j, d
        lane32 = _f32(lane_pts).reshape(-1, 3)
        yaw[vi] = lane32[j, 2]
        for k in vi:
            t[k], heading[k] = box_assemble_waymo(cg[k], pose_inv, PRIORS_WLH[class_id[k]], yaw[k], IS_VEHICLE[class_id[k]])
    keep = np.zeros(n, bool)
    if vi.size:
        keep[vi] = circle_nms(t[vi, 0], t[vi, 1], np.asarray(scores, np.float64)[vi], WAYMO_NMS_GROUP[np.asarray(class_id)[vi]], WAYMO_NMS_THR)
    return dict(valid=valid, keep=keep, translation=t, heading=heading, lane_idx=lane_j, lane_dist=lane_d, yaw=yaw, centroid_global=cg)


# ------------------------------------------------------------------ f4: SAM3D fusion matching
def bev_box(cx, cy, length, width, heading):
    """Box record of orc_bev_iou / orc_bev_match: cos/sin of the heading are taken on the host."""
    return np.array([cx, cy, length, width, np.cos(heading), np.sin(heading)], np.float64)


def bev_iou(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return float(lib().orc_bev_iou(_ptr(a), _ptr(b)))


def bev_match(pred, gt, iou_thr=0.2, want_weights=False):
    """`match(pred, sam3d, iou, TYPE_2D)` of linear_matching.py:53-104 for one sample on (P,6)/(G,6) records:
    returns pred_match (gt index or -1), gt_match, match_iou, total weight[, weights]."""
    pred = np.ascontiguousarray(pred, np.float64).reshape(-1, 6)
    gt = np.ascontiguousarray(gt, np.float64).reshape(-1, 6)
    P, G = pred.shape[0], gt.shape[0]
    pm, gm, iou = np.empty(P, np.int32), np.empty(G, np.int32), np.empty(P, np.float64)
    W = np.zeros((P, G), np.int32)
    total = lib().orc_bev_match(_ptr(pred), P, _ptr(gt), G, float(iou_thr), _ptr(pm), _ptr(gm), _ptr(iou), _ptr(W))
    return (pm, gm, iou, int(total), W) if want_weights else (pm, gm, iou, int(total))
