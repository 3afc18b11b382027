"""Single-call mirrors of the reference's helper functions, each running on the GPU
through the C-ABI (no CPU fallback).  Same names and argument meaning as in the reference's
src/nuscenes/2d_to_3d.py so that parity tests read like calls into the reference:

    get_medoid(points)                                   :116-119
    lane_yaws_distances_and_coords(centroids, lane_pts)  :277-302
    push_centroid(centroid, extents, yaw, poserecord)    :164-198 (+ the rotation of :788-806)
    circle_nms(dets, det_labels, threshs_by_label)       :309-332
    points_in_masks(points, cams, masks, cam_nums)       :553-620 for all masks of a frame
    erode(mask) / decode(rles)                           :526-527 / :425

They are conveniences for tests and small jobs; the batched path is lifting.LiftEngine.
"""
import numpy as np
import torch

from . import _lib
from . import rle as rlemod
from ._lib import check


def _dev():
    if not torch.cuda.is_available():
        raise _lib.Cm3dError("no HIP device: cm3d_amd.ops only runs on the GPU")
    return torch.device("cuda", torch.cuda.current_device())


def _t(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).to(_dev())


def _st():
    return torch.cuda.current_stream().cuda_stream


def _e(*shape, dtype=torch.int32):
    return torch.empty(*shape, dtype=dtype, device=_dev())


def _ws(nbytes):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=_dev())


# ----------------------------------------------------------------------------- masks
def decode(rles, as_counts=False):
    """pycocotools.mask.decode equivalent on the GPU: list of COCO RLE dicts (size [W,H]) ->
    uint8 tensor (n, H, W) (the reference's `depth_images` after its transpose, :425-428)."""
    L = _lib.lib()
    W, H = rles[0]["size"]
    cnts = [rlemod.string_to_counts(r["counts"]) if not as_counts else np.asarray(r["counts"], np.uint32) for r in rles]
    off = np.concatenate([[0], np.cumsum([c.size for c in cnts])]).astype(np.int32)
    allc = np.concatenate(cnts).astype(np.uint32)
    d_c, d_o = _t(allc.view(np.int32)), _t(off)
    dense = _e(len(rles), H, W, dtype=torch.uint8)
    ws = _ws(L.cm3d_rle_workspace_bytes(allc.size))
    check(L.cm3d_rle_to_dense(d_c.data_ptr(), d_o.data_ptr(), len(rles), allc.size, W, H, dense.data_ptr(), ws.data_ptr(),
                              ws.numel(), _st()), "cm3d_rle_to_dense")
    return dense


def erode(dense_masks):
    """cv2.erode(mask, ones((3,3))) for a stack of masks, bit-packed: returns (packed (n,H,Wp) int32, bbox (n,8): bounds of the eroded
    pixels [0..3] and the rectangle `packed` stores [4..7] -- here the whole image; unpack_bits turns both into pixels)."""
    L = _lib.lib()
    d = dense_masks if torch.is_tensor(dense_masks) else _t(dense_masks, np.uint8)
    n, H, W = d.shape
    packed = _e(n, H, (W + 31) // 32)
    bbox = _e(n, _lib.BBOX_STRIDE)
    check(L.cm3d_erode_pack(d.data_ptr(), n, W, H, packed.data_ptr(), bbox.data_ptr(), _st()), "cm3d_erode_pack")
    return packed, bbox


def erode_rle(rle_counts_list, W, H):
    """Same result as erode(decode(.)) straight from run lengths (f1)."""
    L = _lib.lib()
    off = np.concatenate([[0], np.cumsum([len(c) for c in rle_counts_list])]).astype(np.int32)
    allc = np.concatenate([np.asarray(c, np.uint32) for c in rle_counts_list]).astype(np.uint32)
    n = len(rle_counts_list)
    d_c, d_o = _t(allc.view(np.int32)), _t(off)
    packed = torch.zeros(n, H, (W + 31) // 32, dtype=torch.int32, device=_dev())
    bbox = _e(n, _lib.BBOX_STRIDE)
    ws = _ws(L.cm3d_rle_workspace_bytes(allc.size))
    check(L.cm3d_rle_erode_pack(d_c.data_ptr(), d_o.data_ptr(), n, allc.size, W, H, packed.data_ptr(), bbox.data_ptr(),
                                ws.data_ptr(), ws.numel(), _st()), "cm3d_rle_erode_pack")
    return packed, bbox


def unpack_bits(packed, W, bbox=None):
    """(n,H,Wp) int32 bit-packed -> (n,H,W) uint8 numpy (host helper for tests/tools).  bbox (n,8) from erode / erode_rle: a
    mask's slot holds the rows of its stored rectangle (xw0, y0, wc, rows = bbox[:, 4:8]) one behind the other
    (include/cm3d_hip.h); without it the slots are read as whole images."""
    p = packed.cpu().numpy().view(np.uint32)
    n, H, Wp = p.shape
    if bbox is not None:
        rc = (bbox.cpu().numpy() if torch.is_tensor(bbox) else np.asarray(bbox))[:, 4:8]
        full = np.zeros_like(p)
        flat = p.reshape(n, -1)
        for i, (xw0, y0, wc, rows) in enumerate(rc.tolist()):
            if wc > 0 and rows > 0:
                full[i, y0:y0 + rows, xw0:xw0 + wc] = flat[i, :rows * wc].reshape(rows, wc)
        p = full
    bits = ((p[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8)
    return bits.reshape(n, H, -1)[:, :, :W]


# ----------------------------------------------------------------------------- a4-a8
def points_in_masks(points, cams, packed, bbox, cam_nums, W, H, min_dist=2.3):
    """All masks of ONE frame.  points (N,4) float32 global frame; cams (C,CAM_STRIDE);
    packed/bbox from erode().  Returns the list of ascending index arrays (track_points, :617)."""
    L = _lib.lib()
    pts = _t(points, np.float32)
    N, n = pts.shape[0], len(cam_nums)
    d_cams = _t(np.asarray(cams, np.float32))
    pt_off, mask_off = _t(np.array([0, N], np.int32)), _t(np.array([0, n], np.int32))
    mask_cam = _t(np.asarray(cam_nums, np.int32))
    planes = (n + 31) // 32
    hit_words, hit_count = _e(planes, N), _e(n)
    status = _e(_lib.STATUS_WORDS)
    hit_off, tile_off = _e(n + 1), _e(n + 1)
    cap = max(1024, N * 8)
    hit_idx = _e(cap)
    st = _st()
    ws = _ws(L.cm3d_project_workspace_bytes(1, N, planes))
    check(L.cm3d_batch_begin(status.data_ptr(), hit_count.data_ptr(), n, 0, 0, st), "cm3d_batch_begin")
    check(L.cm3d_project_hits(pts.data_ptr(), pt_off.data_ptr(), 1, N, N, d_cams.data_ptr(), d_cams.shape[0], mask_off.data_ptr(),
                              mask_cam.data_ptr(), bbox.data_ptr(), packed.data_ptr(), n, W, H, float(np.float32(min_dist)), planes,
                              hit_words.data_ptr(), hit_count.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), 0, 0, st),
          "cm3d_project_hits")
    check(L.cm3d_compact_hits(hit_words.data_ptr(), planes, 1, N, N, mask_off.data_ptr(), n, hit_count.data_ptr(), 0, 0, 0, 0, 0,
                              pts.data_ptr(), hit_off.data_ptr(), tile_off.data_ptr(), hit_idx.data_ptr(), 0, 0, cap, 0,
                              status.data_ptr(), ws.data_ptr(), ws.numel(), st), "cm3d_compact_hits")
    s = status.cpu().numpy()
    if s[0]:
        raise _lib.Cm3dError(f"status {s}")
    off = hit_off.cpu().numpy()
    idx = hit_idx[: off[-1]].cpu().numpy()
    return [idx[off[i]:off[i + 1]] for i in range(n)]


# ----------------------------------------------------------------------------- a9
def get_medoid(points, want_colsum=False, via_rows=False):
    """points: (3,M) float32 like the reference's argument; returns the medoid column index.  via_rows: hand the
    kernel a cloud + row-index list (its gather form) instead of the contiguous per-hit coordinates."""
    L = _lib.lib()
    p = np.asarray(points.cpu() if torch.is_tensor(points) else points, np.float32)
    M = p.shape[1]
    P4 = np.zeros((M, 4), np.float32)
    P4[:, :3] = p[:3].T
    pts = _t(P4)
    pt_off, mask_frame = _t(np.array([0, M], np.int32)), _t(np.array([0], np.int32))
    ntile = (M + _lib.MEDOID_TILE - 1) // _lib.MEDOID_TILE
    hit_off, tile_off = _t(np.array([0, M], np.int32)), _t(np.array([0, ntile], np.int32))
    hit_idx = _t(np.arange(M, dtype=np.int32))
    med, cen = _e(1), _e(1, 3, dtype=torch.float32)
    colsum = _e(max(M, 1), dtype=torch.float32) if want_colsum else None      # None: lists of > 512 points take the two-pass route
    ws = _ws(L.cm3d_medoid_workspace_bytes(1, max(M, 1)))
    check(L.cm3d_medoid(pts.data_ptr(), pt_off.data_ptr() if via_rows else 0, mask_frame.data_ptr() if via_rows else 0, 1,
                        hit_off.data_ptr(), tile_off.data_ptr(), hit_idx.data_ptr() if via_rows else 0, max(M, 1), 0, med.data_ptr(),
                        cen.data_ptr(), colsum.data_ptr() if want_colsum else 0, ws.data_ptr(), ws.numel(), _st()), "cm3d_medoid")
    j = int(med.cpu()[0])
    return (j, colsum.cpu().numpy()[:M]) if want_colsum else j


def get_medoids(point_lists, via_rows=False, want_colsum=False):
    """Several lists in ONE call of the medoid stage (one "mask" each; `get_medoid` is the reference's one-list signature): what the
    lists of a batch do to each other -- the two-pass route is taken by lists of more than 256 points when the batch holds one of
    more than 448 (csrc/medoid.hip) -- can only be tested this way.  point_lists: arrays (M_k, 3); returns the list of medoid indices
    (and the exact column sums with want_colsum, which also forces the one-pass route)."""
    L = _lib.lib()
    Ms = [int(np.asarray(p).shape[0]) for p in point_lists]
    n, tot = len(Ms), int(sum(Ms))
    P4 = np.zeros((max(tot, 1), 4), np.float32)
    if tot:
        P4[:tot, :3] = np.concatenate([np.asarray(p, np.float32).reshape(-1, 3) for p in point_lists], 0)
    hit_off = np.concatenate([[0], np.cumsum(Ms)]).astype(np.int32)
    tiles = [(m + _lib.MEDOID_TILE - 1) // _lib.MEDOID_TILE for m in Ms]
    tile_off = np.concatenate([[0], np.cumsum(tiles)]).astype(np.int32)
    pts = _t(P4)
    # gather form: every list is the cloud of a frame of its own, its rows listed in order
    pt_off, mask_frame = _t(hit_off.copy()), _t(np.arange(n, dtype=np.int32))
    hit_idx = _t(np.concatenate([np.arange(m, dtype=np.int32) for m in Ms]) if tot else np.zeros(1, np.int32))
    d_hit_off, d_tile_off = _t(hit_off), _t(tile_off)
    med, cen = _e(n), _e(n, 3, dtype=torch.float32)
    colsum = _e(max(tot, 1), dtype=torch.float32) if want_colsum else None
    ws = _ws(L.cm3d_medoid_workspace_bytes(n, max(tot, 1)))
    check(L.cm3d_medoid(pts.data_ptr(), pt_off.data_ptr() if via_rows else 0, mask_frame.data_ptr() if via_rows else 0, n,
                        d_hit_off.data_ptr(), d_tile_off.data_ptr(), hit_idx.data_ptr() if via_rows else 0, max(tot, 1), 0, med.data_ptr(),
                        cen.data_ptr(), colsum.data_ptr() if want_colsum else 0, ws.data_ptr(), ws.numel(), _st()), "cm3d_medoid")
    out = [int(v) for v in med.cpu().numpy()]
    if want_colsum:
        cs = colsum.cpu().numpy()
        return out, [cs[hit_off[k]:hit_off[k + 1]] for k in range(n)]
    return out


# ----------------------------------------------------------------------------- a10
def lane_yaws_distances_and_coords(all_centroids, all_lane_pts):
    """Returns (yaws, distances, coords) like the reference; yaws/coords are float32-valued."""
    L = _lib.lib()
    cent = np.asarray(all_centroids, np.float64).astype(np.float32).reshape(-1, 3)
    lane = np.asarray(all_lane_pts, np.float64).astype(np.float32).reshape(-1, 3)
    K = cent.shape[0]
    d_c, d_l = _t(cent), _t(lane)
    med = _t(np.zeros(K, np.int32))
    mask_frame = _t(np.zeros(K, np.int32))
    lane_off, frame_lane = _t(np.array([0, lane.shape[0]], np.int32)), _t(np.array([0], np.int32))
    idx, dist = _e(K), _e(K, dtype=torch.float64)
    ws = _ws(L.cm3d_lane_nn_workspace_bytes(K))
    grid = _ws(L.cm3d_lane_grid_bytes(1, lane.shape[0]))
    check(L.cm3d_lane_grid_build(d_l.data_ptr(), lane_off.data_ptr(), 1, lane.shape[0], grid.data_ptr(), grid.numel(), _st()),
          "cm3d_lane_grid_build")
    check(L.cm3d_lane_nn(d_c.data_ptr(), med.data_ptr(), mask_frame.data_ptr(), K, d_l.data_ptr(), lane_off.data_ptr(),
                         frame_lane.data_ptr(), 1, lane.shape[0], grid.data_ptr(), idx.data_ptr(), dist.data_ptr(), ws.data_ptr(),
                         ws.numel(), _st()), "cm3d_lane_nn")
    j = idx.cpu().numpy()
    return lane[j, 2], dist.cpu().numpy(), lane[j, :2]


# ----------------------------------------------------------------------------- a11-a15
def _box_nms(centroids, class_ids, scores, yaws, ego_xyz, classes, valid=None):
    """One frame through cm3d_box_nms with given lane yaws (a one-point lane table per mask)."""
    L = _lib.lib()
    cent = np.asarray(centroids, np.float32).reshape(-1, 3)
    n = cent.shape[0]
    lane = np.zeros((n, 3), np.float32)
    lane[:, 2] = np.asarray(yaws, np.float32)
    med = np.zeros(n, np.int32) if valid is None else np.where(np.asarray(valid, bool), 0, -1).astype(np.int32)
    d = dict(cent=_t(cent), med=_t(med), mask_off=_t(np.array([0, n], np.int32)), cls=_t(np.asarray(class_ids, np.int32)),
             score=_t(np.asarray(scores, np.float64)), lane=_t(lane), lane_off=_t(np.array([0, n], np.int32)),
             frame_lane=_t(np.array([0], np.int32)), lane_idx=_t(np.arange(n, dtype=np.int32)), lane_dist=_t(np.zeros(n, np.float64)),
             prior=_t(classes.prior_wlh), veh=_t(classes.is_vehicle), thr=_t(classes.nms_thr),
             grp=_t(np.ascontiguousarray(classes.nms_group, np.int32)), ego=_t(np.asarray(ego_xyz, np.float64).reshape(1, 3)))
    box, flags = _e(n, _lib.BOX_STRIDE, dtype=torch.float64), _e(n)
    check(L.cm3d_box_nms(d["cent"].data_ptr(), d["med"].data_ptr(), d["mask_off"].data_ptr(), 1, n, d["cls"].data_ptr(),
                         d["score"].data_ptr(), d["lane"].data_ptr(), d["lane_off"].data_ptr(), d["frame_lane"].data_ptr(),
                         d["lane_idx"].data_ptr(), d["lane_dist"].data_ptr(), d["prior"].data_ptr(), d["veh"].data_ptr(),
                         d["grp"].data_ptr(), d["thr"].data_ptr(), len(classes.names), d["ego"].data_ptr(), 0, box.data_ptr(),
                         flags.data_ptr(), _st()),
          "cm3d_box_nms")
    return box.cpu().numpy(), flags.cpu().numpy()


def push_centroid(centroid, class_name, lane_yaw, poserecord, classes=None):
    """Returns (pushed_centroid (3,), rotation wxyz (4,)) for one box of a pushed class."""
    from .lifting import ClassTable
    classes = classes or ClassTable.nuscenes()
    box, _ = _box_nms([centroid], [classes.index(class_name)], [1.0], [lane_yaw], poserecord["translation"], classes)
    return box[0, 0:3].copy(), np.array([box[0, 3], 0.0, 0.0, box[0, 4]])


def circle_nms(dets, det_labels, threshs_by_label):
    """dets (n,3) x,y,score; det_labels list of class names; returns the kept indices (ascending),
    i.e. sorted(reference circle_nms(...))."""
    L = _lib.lib()
    names = list(threshs_by_label.keys())
    dets = np.asarray(dets, np.float64).reshape(-1, 3)
    n = dets.shape[0]
    if n == 0:
        return []
    x, y, sc = _t(dets[:, 0].copy()), _t(dets[:, 1].copy()), _t(dets[:, 2].copy())
    lab = _t(np.array([names.index(l) for l in det_labels], np.int32))
    thr = _t(np.array([threshs_by_label[k] for k in names], np.float64))
    off = _t(np.array([0, n], np.int32))
    keep = _e(n)
    check(L.cm3d_circle_nms(x.data_ptr(), y.data_ptr(), sc.data_ptr(), lab.data_ptr(), off.data_ptr(), 1, thr.data_ptr(),
                            len(names), keep.data_ptr(), _st()), "cm3d_circle_nms")
    return np.flatnonzero(keep.cpu().numpy()).tolist()


# ----------------------------------------------------------------------------- f4: SAM3D fusion matching
def match_records(boxes7):
    """(n,7) [center_x, center_y, bottom_z, length, width, height, heading] -> (n,6) float64 records of
    cm3d_bev_match.  The values pass through float32 like `tf.convert_to_tensor(boxes, dtype=float)`
    (linear_matching.py:248-249); cos/sin of the float32 heading are taken in float64 on the host."""
    b = np.asarray(boxes7, np.float64).reshape(-1, 7).astype(np.float32).astype(np.float64)
    return np.stack([b[:, 0], b[:, 1], b[:, 3], b[:, 4], np.cos(b[:, 6]), np.sin(b[:, 6])], axis=1)


def bev_match(pred_boxes, gt_boxes, iou=0.2):
    """`match(pred_boxes, sam3d_boxes, iou, Type.TYPE_2D)` (linear_matching.py:53-104) for a list of samples in one
    GPU call.  pred_boxes / gt_boxes: lists (one entry per sample) of (n,7) arrays.  Returns one
    (prediction_ids, groundtruth_ids, ious) triple per sample, matches in ascending prediction order."""
    L = _lib.lib()
    F = len(pred_boxes)
    if F == 0:
        return []
    if len(gt_boxes) != F:
        raise ValueError("bev_match: one gt entry per sample")
    pr = [match_records(b) for b in pred_boxes]
    gr = [match_records(b) for b in gt_boxes]
    np_, ng = np.array([r.shape[0] for r in pr], np.int64), np.array([r.shape[0] for r in gr], np.int64)
    if max(np_.max(), ng.max()) > _lib.MAX_MATCH_BOXES:
        raise _lib.Cm3dError(f"bev_match: more than {_lib.MAX_MATCH_BOXES} boxes in a sample")
    p_off = np.concatenate([[0], np.cumsum(np_)]).astype(np.int32)
    g_off = np.concatenate([[0], np.cumsum(ng)]).astype(np.int32)
    pair_off = np.concatenate([[0], np.cumsum(np_ * ng)]).astype(np.int64)
    n_pred, n_gt, total = int(p_off[-1]), int(g_off[-1]), int(pair_off[-1])
    d_p = _t(np.concatenate(pr).reshape(-1, 6) if n_pred else np.zeros((1, 6)), np.float64)
    d_g = _t(np.concatenate(gr).reshape(-1, 6) if n_gt else np.zeros((1, 6)), np.float64)
    d_po, d_go, d_pair = _t(p_off), _t(g_off), _t(pair_off)
    pm, gm = _e(max(n_pred, 1)), _e(max(n_gt, 1))
    miou = _e(max(n_pred, 1), dtype=torch.float64)
    status = torch.zeros(1, dtype=torch.int32, device=_dev())
    ws = _ws(L.cm3d_bev_match_workspace_bytes(total))
    check(L.cm3d_bev_match(d_p.data_ptr(), d_po.data_ptr(), n_pred, d_g.data_ptr(), d_go.data_ptr(), n_gt, d_pair.data_ptr(), F,
                           total, float(iou), pm.data_ptr(), gm.data_ptr(), miou.data_ptr(), status.data_ptr(), ws.data_ptr(),
                           ws.numel(), _st()), "cm3d_bev_match")
    pm, miou = pm.cpu().numpy()[:n_pred], miou.cpu().numpy()[:n_pred]
    if int(status.item()) != 0:
        raise _lib.Cm3dError("cm3d_bev_match: a sample exceeded the box capacity")
    out = []
    for f in range(F):
        m = pm[p_off[f]:p_off[f + 1]]
        ids = np.flatnonzero(m >= 0)
        out.append((ids.astype(np.int64), m[ids].astype(np.int64), miou[p_off[f]:p_off[f + 1]][ids]))
    return out
