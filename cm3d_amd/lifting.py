"""Host-side driver of the lifting path: packs frames into a lift batch, keeps
the device buffers (PyTorch-ROCm tensors are used for device memory and streams
only) and calls the HIP kernels through the C-ABI of include/cm3d_hip.h.

Mirrors the stages of the reference's src/nuscenes/2d_to_3d.py main loop
(:415-694 stage 1, :699-827 stage 2, :844-924 NMS) for a whole batch of frames
at once; names follow the reference (`get_detection_name`, `ATTRIBUTE_NAMES`,
`threshs_by_label`, ...).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import os
import zlib

import numpy as np
import torch

from . import _lib
from . import rle as rlemod
from ._lib import Cm3dError, check

# ---- tables of the reference --------------------------------------------------
CAM_LIST = ["CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT", "CAM_FRONT_LEFT"]  # :62-69
ATTRIBUTE_NAMES = {  # :70-81
    "barrier": "", "traffic_cone": "", "bicycle": "cycle.without_rider", "motorcycle": "cycle.without_rider",
    "pedestrian": "pedestrian.standing", "car": "vehicle.stopped", "bus": "vehicle.stopped",
    "construction_vehicle": "vehicle.stopped", "trailer": "vehicle.stopped", "truck": "vehicle.stopped",
}
SHAPE_PRIORS_CHATGPT = {  # cfg/shape_priors_chatgpt.json:1-12, [w, l, h]
    "car": [1.8, 4.5, 1.4], "truck": [2.6, 8.0, 3.6], "bus": [2.5, 12.0, 4.0], "trailer": [2.6, 12.0, 3.6],
    "construction_vehicle": [2.0, 4.5, 2.5], "pedestrian": [0.4, 0.7, 1.7], "motorcycle": [0.8, 2.1, 1.7],
    "bicycle": [0.6, 1.8, 1.4], "traffic_cone": [0.3, 0.3, 0.7], "barrier": [0.5, 1.2, 0.9],
}
THRESHS_BY_LABEL = {  # :850-861 (squared metres)
    "barrier": 1, "traffic_cone": 0.175, "bicycle": 0.85, "motorcycle": 0.85, "pedestrian": 0.175, "car": 4,
    "bus": 10, "construction_vehicle": 12, "trailer": 10, "truck": 12,
}
PUSHED_CLASSES = ["car", "truck", "bus", "construction_vehicle", "trailer", "barrier"]  # :763
MIN_DIST = 2.3            # :348
N_SWEEPS = 3              # :437


def get_detection_name(name):
    """reference :122-132"""
    return {"trafficcone": "traffic_cone", "constructionvehicle": "construction_vehicle", "human": "pedestrian"}.get(name, name)


def get_shape_prior(shape_priors, name, chatgpt=True):
    """reference :134-161 (only the chatgpt table is ever loaded, :384-385)"""
    if not chatgpt:
        raise NotImplementedError("the reference only ships the chatgpt priors on this path")
    return shape_priors[name]


@dataclass
class ClassTable:
    names: List[str]
    prior_wlh: np.ndarray      # (n,3) float64
    is_vehicle: np.ndarray     # (n,) int32
    nms_thr: np.ndarray        # (g,) float64, squared-distance threshold per NMS label
    nms_group: Optional[np.ndarray] = None   # (n,) int32 NMS label of each class (default: the class itself)
    out_names: Optional[List[str]] = None    # what the writer calls each class (Waymo: the Waymo type)

    def __post_init__(self):
        if self.nms_group is None:
            self.nms_group = np.arange(len(self.names), dtype=np.int32)

    @staticmethod
    def nuscenes(shape_priors=None):
        pri = shape_priors or SHAPE_PRIORS_CHATGPT
        names = list(pri.keys())
        return ClassTable(names=names,
                          prior_wlh=np.array([pri[n] for n in names], np.float64),
                          is_vehicle=np.array([n in PUSHED_CLASSES for n in names], np.int32),
                          nms_thr=np.array([THRESHS_BY_LABEL[n] for n in names], np.float64))

    @staticmethod
    def waymo(shape_priors=None):
        """Reference src/waymo: same priors and pushed classes, classes renamed through NUSC_TO_WAYMO
        (cfg/prompt_cfg.py:286-297), NMS per Waymo type with the thresholds of 2d_to_3d.py:1147-1158."""
        from . import waymo as wm
        base = ClassTable.nuscenes(shape_priors)
        types = [wm.WAYMO_TYPE.get(wm.NUSC_TO_WAYMO[n], -1) for n in base.names]
        thr = np.zeros(5, np.float64)
        for t, v in wm.THRESHS_BY_TYPE.items():
            thr[t] = v
        return ClassTable(names=base.names, prior_wlh=base.prior_wlh, is_vehicle=base.is_vehicle, nms_thr=thr,
                          nms_group=np.array([max(t, 0) for t in types], np.int32),
                          out_names=[wm.NUSC_TO_WAYMO[n] for n in base.names])

    def index(self, detection_name):
        return self.names.index(detection_name)   # ValueError for an unknown class, like the reference's KeyError


# ---- batch packing ------------------------------------------------------------
@dataclass
class HostBatch:
    """numpy view of one lift batch (see include/cm3d_hip.h for the layout)."""
    raw: np.ndarray
    raw_stride: int
    sweep_row_off: np.ndarray
    sweep_xf: np.ndarray
    frame_sweep_off: np.ndarray
    max_rows_per_sweep: int
    cams: np.ndarray
    n_cams: int
    mask_off: np.ndarray
    mask_cam: np.ndarray
    mask_frame: np.ndarray
    rle_counts: np.ndarray
    rle_off: np.ndarray
    class_id: np.ndarray
    score: np.ndarray
    lane: np.ndarray
    lane_off: np.ndarray
    frame_lane: np.ndarray
    ego_xyz: np.ndarray
    width: int
    height: int
    tokens: List[str]
    labels: List[List[str]]
    intensity: Optional[np.ndarray] = None    # quad layout only: (rows,) float32, the rows' fourth column (None: not uploaded)
    frame_rows: Optional[np.ndarray] = None   # quad layout only: (F,) rows of each frame without its padding rows
    pose_rt: Optional[np.ndarray] = None      # (F,12) float32, Waymo: vehicle -> global rotate/translate
    pose_inv: Optional[np.ndarray] = None     # (F,16) float32, Waymo: inverse of the float32 frame pose
    ego_box: bool = True                      # nuScenes drops the ego-box points (:442-445); Waymo does not

    @property
    def n_frames(self):
        return len(self.tokens)

    @property
    def n_masks(self):
        return int(self.mask_off[-1])

    @property
    def n_raw_rows(self):
        return int(self.sweep_row_off[-1])

    @property
    def quads(self):
        return self.raw_stride == _lib.RAW_QUADS

    @property
    def n_real_rows(self):
        """Rows the files held (the quad layout's padding rows left out)."""
        return self.n_raw_rows if self.frame_rows is None else int(np.sum(self.frame_rows))

    @property
    def bytes_per_row(self):
        """Bytes of a raw row that cross HBM in the projection launch."""
        return 12 if self.quads else 4 * self.raw_stride


def rows_to_quads(raw, sweep_row_off, frame_sweep_off, keep_intensity=True):
    """Row-major sweeps (rows, >= 4 columns) -> the quad layout of include/cm3d_hip.h (cm3d_sweep_prep): every frame padded to a
    multiple of 4 rows with NaN rows that belong to its last sweep, rows 4q..4q+3 stored as x0..3 y0..3 z0..3.
    Returns (quads (R/4, 3, 4) float32, intensity (R,) float32 or None, sweep_row_off' (S+1,), frame_rows (F,)).
    What the native reader does while it fills the page-locked batch (cm3d_amd.reader); this is the numpy form for
    batches packed from arrays."""
    raw = np.asarray(raw, np.float32)
    sro = np.asarray(sweep_row_off, np.int64)
    fso = np.asarray(frame_sweep_off, np.int64)
    F = len(fso) - 1
    frame_rows = (sro[fso[1:]] - sro[fso[:-1]]).astype(np.int32)
    padded = (frame_rows.astype(np.int64) + 3) // 4 * 4
    new_start = np.concatenate([[0], np.cumsum(padded)])
    R = int(new_start[-1])
    xyz = np.full((R, 3), np.nan, np.float32)
    inten = np.zeros(R, np.float32) if keep_intensity else None
    new_sro = np.zeros(len(sro), np.int64)
    for f in range(F):
        a, b = int(sro[fso[f]]), int(sro[fso[f + 1]])
        d = int(new_start[f]) - a
        xyz[a + d:b + d] = raw[a:b, :3]
        if keep_intensity:
            inten[a + d:b + d] = raw[a:b, 3]
        new_sro[fso[f]:fso[f + 1]] = sro[fso[f]:fso[f + 1]] + d
    new_sro[fso[-1]:] = R                      # the end of the last sweep = the padded end of the batch
    # sweeps of a frame stay back to back; a frame's padding rows sit behind its last sweep, in front of the next frame's first
    quads = np.ascontiguousarray(xyz.reshape(R // 4, 4, 3).transpose(0, 2, 1))
    return quads, inten, new_sro.astype(np.int32), frame_rows


def default_layout():
    """Layout of the raw rows in a packed batch: "quads" (12 bytes per row cross HBM; include/cm3d_hip.h) unless
    CM3D_RAW_LAYOUT=rows asks for the files' own rows."""
    return os.environ.get("CM3D_RAW_LAYOUT", "quads")


def pack_frames(frames: Sequence, lane_tables: Sequence[np.ndarray], frame_lane: Sequence[int],
                classes: Optional[ClassTable] = None, layout: Optional[str] = None, keep_intensity: bool = True) -> HostBatch:
    """frames: objects with the attributes of cm3d_amd.synthetic.Frame.  layout: "rows" (the sweeps as they are) or
    "quads" (rows_to_quads); None = default_layout()."""
    classes = classes or ClassTable.nuscenes()
    layout = layout or default_layout()
    if layout not in ("rows", "quads"):
        raise ValueError(layout)
    W, H = frames[0].width, frames[0].height
    n_cams = frames[0].cams.shape[0]
    raws, xfs, row_off, fso = [], [], [0], [0]
    cams, mask_off, mask_cam, mask_frame = [], [0], [], []
    cnts, rle_off, class_id, score, ego, tokens, labels = [], [0], [], [], [], [], []
    pose_rt, pose_inv = [], []
    stride = frames[0].sweeps_raw[0].shape[1]
    for fi, fr in enumerate(frames):
        if fr.width != W or fr.height != H or fr.cams.shape[0] != n_cams:
            raise ValueError("all frames of a batch must share mask size and camera count")
        for r in fr.sweeps_raw:
            r = np.ascontiguousarray(r, np.float32)
            if r.shape[1] != stride:
                raise ValueError("mixed sweep strides")
            raws.append(r)
            row_off.append(row_off[-1] + r.shape[0])
        xfs.append(np.asarray(fr.sweep_xf, np.float32).reshape(-1, _lib.SWEEP_XF_STRIDE))
        fso.append(fso[-1] + len(fr.sweeps_raw))
        cams.append(np.asarray(fr.cams, np.float32))
        n = len(fr.rles)
        if not (len(fr.labels) == len(fr.scores) == len(fr.cam_nums) == n):
            raise ValueError("labels / detection_scores / cam_nums / masks differ in length")
        for rl in fr.rles:
            if list(rl["size"]) != [W, H]:
                raise ValueError(f"mask size {rl['size']} != [{W},{H}]")
            c = rlemod.string_to_counts(rl["counts"]) if isinstance(rl["counts"], (bytes, bytearray)) else np.asarray(rl["counts"], np.uint32)
            if int(c.astype(np.int64).sum()) != W * H:
                raise ValueError("RLE run lengths do not cover the mask")
            cnts.append(c)
            rle_off.append(rle_off[-1] + c.size)
        mask_off.append(mask_off[-1] + n)
        mask_cam.extend(int(c) for c in fr.cam_nums)
        mask_frame.extend([fi] * n)
        for l in fr.labels:
            ci = classes.index(get_detection_name(l))
            if classes.out_names is not None and classes.out_names[ci] == "":
                raise ValueError(f"label {l!r} has no output type")     # the reference raises ValueError (src/waymo/2d_to_3d.py:1054-1062)
            class_id.append(ci)
        score.extend(float(s) for s in fr.scores)
        ego.append(np.asarray(fr.ego_xyz, np.float64))
        tokens.append(fr.token)
        labels.append(list(fr.labels))
        if getattr(fr, "pose", None) is not None:
            from . import waymo as wm
            rt, inv = wm.pose_records(fr.pose)
            pose_rt.append(rt)
            pose_inv.append(inv)
    lane32 = [np.asarray(t, np.float64).astype(np.float32).reshape(-1, 3) for t in lane_tables]   # torch.Tensor(...) at :278
    lane_off = np.concatenate([[0], np.cumsum([t.shape[0] for t in lane32])]).astype(np.int32)
    i32 = lambda a: np.asarray(a, np.int32)
    raw = np.concatenate(raws, 0) if raws else np.zeros((0, stride), np.float32)
    intensity = frame_rows = None
    row_off = i32(row_off)
    if layout == "quads":
        raw, intensity, row_off, frame_rows = rows_to_quads(raw, row_off, fso, keep_intensity)
        stride = _lib.RAW_QUADS
    return HostBatch(
        raw=raw, raw_stride=stride, intensity=intensity, frame_rows=frame_rows,
        sweep_row_off=row_off, sweep_xf=np.concatenate(xfs, 0), frame_sweep_off=i32(fso),
        max_rows_per_sweep=max(1, int(np.diff(row_off).max())) if len(row_off) > 1 else 1, cams=np.stack(cams), n_cams=n_cams,
        mask_off=i32(mask_off), mask_cam=i32(mask_cam), mask_frame=i32(mask_frame),
        rle_counts=np.concatenate(cnts).astype(np.uint32) if cnts else np.zeros(0, np.uint32), rle_off=i32(rle_off),
        class_id=i32(class_id), score=np.asarray(score, np.float64),
        lane=np.concatenate(lane32, 0), lane_off=lane_off, frame_lane=i32(frame_lane),
        ego_xyz=np.stack(ego), width=W, height=H, tokens=tokens, labels=labels,
        pose_rt=np.stack(pose_rt).astype(np.float32) if len(pose_rt) == len(frames) and frames else None,
        pose_inv=np.stack(pose_inv).astype(np.float32) if len(pose_inv) == len(frames) and frames else None,
        ego_box=not ((len(pose_rt) == len(frames) and len(frames) > 0) or all(getattr(f, "no_ego_box", False) for f in frames)))


def pack_manifest(man, lane_tables, frame_lane, classes: Optional[ClassTable], rd, stride=5, alloc=None):
    """Frame manifests (nusc_io.scene_manifest) -> HostBatch, with the bulk data read by the native loader `rd`
    (cm3d_amd.reader.Reader): all sweeps of the batch land in one (page-locked) `raw` buffer, all RLE strings are parsed
    to run lengths by its thread pool.  Frames without masks are left out, like pack_frames' callers do.
    Returns (HostBatch or None when no frame has a mask, indices of the frames it holds)."""
    classes = classes or ClassTable.nuscenes()
    counts, rle_off, fmo, wh = rd.load_masks([m.mask_path for m in man])
    n_per = np.diff(fmo)
    live = [i for i in range(len(man)) if n_per[i] > 0]
    if not live:
        return None, []
    if len(live) != len(man):                       # rare: repack the frames that have masks
        hb, sub = pack_manifest([man[i] for i in live], lane_tables, [frame_lane[i] for i in live], classes, rd, stride, alloc)
        return hb, [live[k] for k in sub]
    W, H = int(wh[0, 0]), int(wh[0, 1])
    if np.any(wh[:, 0] != W) or np.any(wh[:, 1] != H):
        raise ValueError("all frames of a batch must share mask size and camera count")
    fso = np.concatenate([[0], np.cumsum([len(m.sweep_paths) for m in man])]).astype(np.int32)
    intensity = frame_rows = None
    if default_layout() == "quads":     # the files' rows go straight into the quad layout (no intensity plane: no output holds it)
        raw, intensity, row_off, frame_rows = rd.load_sweeps_quads([p for m in man for p in m.sweep_paths], fso, stride, False, alloc)
        stride = _lib.RAW_QUADS
    else:
        raw, row_off = rd.load_sweeps([p for m in man for p in m.sweep_paths], stride, alloc)
    n_cams = man[0].cams.shape[0]
    mask_cam, mask_frame, class_id, score = [], [], [], []
    for fi, m in enumerate(man):
        n = int(n_per[fi])
        if not (len(m.labels) == len(m.scores) == len(m.cam_nums) == n):
            raise ValueError("labels / detection_scores / cam_nums / masks differ in length")
        if m.cams.shape[0] != n_cams:
            raise ValueError("all frames of a batch must share mask size and camera count")
        mask_cam.extend(int(c) for c in m.cam_nums)
        mask_frame.extend([fi] * n)
        for l in m.labels:
            ci = classes.index(get_detection_name(l))
            if classes.out_names is not None and classes.out_names[ci] == "":
                raise ValueError(f"label {l!r} has no output type")
            class_id.append(ci)
        score.extend(float(s) for s in m.scores)
    lane32 = [np.asarray(t, np.float64).astype(np.float32).reshape(-1, 3) for t in lane_tables]
    lane_off = np.concatenate([[0], np.cumsum([t.shape[0] for t in lane32])]).astype(np.int32)
    i32 = lambda a: np.asarray(a, np.int32)
    hb = HostBatch(
        raw=raw, raw_stride=stride, intensity=intensity, frame_rows=frame_rows, sweep_row_off=row_off,
        sweep_xf=np.concatenate([np.asarray(m.sweep_xf, np.float32).reshape(-1, _lib.SWEEP_XF_STRIDE) for m in man], 0),
        frame_sweep_off=fso, max_rows_per_sweep=max(1, int(np.diff(row_off).max())), cams=np.stack([np.asarray(m.cams, np.float32) for m in man]),
        n_cams=n_cams, mask_off=fmo.astype(np.int32), mask_cam=i32(mask_cam), mask_frame=i32(mask_frame), rle_counts=counts, rle_off=rle_off,
        class_id=i32(class_id), score=np.asarray(score, np.float64), lane=np.concatenate(lane32, 0), lane_off=lane_off,
        frame_lane=i32(frame_lane), ego_xyz=np.stack([m.ego_xyz for m in man]), width=W, height=H, tokens=[m.token for m in man],
        labels=[list(m.labels) for m in man], ego_box=True)
    return hb, list(range(len(man)))


# ---- device side ----------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return 0 if t is None else t.data_ptr()


_MFMA_VERIFIED = {}


def _verify_matrix_pipe(lib, dev):
    """Once per process and device: the medoid's first pass over long lists runs on the matrix pipe, and the error bound of
    its second pass holds only while v_mfma_f32_32x32x2_f32 reproduces the reference's k-ordered fmaf chain bit for bit
    (csrc/medoid.hip).  cm3d_selftest_mfma checks exactly that on this device (2048 waves x 24 tiles, coordinates from
    vehicle-frame to 20x nuScenes' global magnitudes, ~0.1 ms); a device that fails it must not produce labels."""
    key = str(dev)
    if key in _MFMA_VERIFIED:
        return
    with torch.cuda.device(dev):
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        check(lib.cm3d_selftest_mfma(20240607, 24, bad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "cm3d_selftest_mfma")
        n_bad = int(bad.item())
    if n_bad:
        raise Cm3dError(f"matrix-pipe self-test failed on {key}: {n_bad} of {2048 * 24 * 1024} squared distances differ from the "
                        "vector fma chain or an output clamp is not honoured; rebuild with -DMD_APPROX_MFMA=0 -DMD_SCALED_ROUTES=0")
    _MFMA_VERIFIED[key] = True


class LiftEngine:
    """Owns the device buffers of one lift batch and launches the kernels.
    All launches go to torch's current HIP stream and never synchronise."""

    def __init__(self, device="cuda:0", classes: Optional[ClassTable] = None, min_dist=MIN_DIST,
                 hits_per_point=4.0, keep_colsum=False, keep_cloud=None, lane_cache=None):
        """keep_cloud: also materialise the transformed cloud (`points`, the reference's aggregated `pc`).  The path
        itself does not need it -- the in-mask points are re-derived from the raw rows (hit_xyz) -- so the default is
        off (CM3D_KEEP_CLOUD=1 turns it on); tests and callers that want the cloud back ask for it."""
        self.lib = _lib.lib()                      # raises when the extension is not built
        if not torch.cuda.is_available():
            raise Cm3dError("no HIP device: the lifting path only runs on the GPU (no CPU fallback)")
        self.dev = torch.device(device)
        self.classes = classes or ClassTable.nuscenes()
        self.min_dist = float(np.float32(min_dist))
        self.halfw = float(np.float32(np.sqrt(min_dist)))       # :443-444
        self.hits_per_point = hits_per_point
        self.keep_colsum = keep_colsum
        self.keep_cloud = (os.environ.get("CM3D_KEEP_CLOUD", "0") == "1") if keep_cloud is None else bool(keep_cloud)
        _verify_matrix_pipe(self.lib, self.dev)
        self.b = None
        self._lane = None                                    # the lane tables on the device and their spatial index (upload)
        self._lane_cache = lane_cache if lane_cache is not None else {}      # table set -> tables + index; a LiftPipeline hands its engines one
        d = self.dev
        self.side = torch.cuda.Stream(device=d)              # lane-grid build overlaps the point/mask stages
        self.fused_sweeps = os.environ.get("CM3D_FUSED_SWEEPS", "1") == "1"    # 0: separate sweep and projection launches
        # the medoid stage's feedback word (stage_medoid): starts at 1 = "expect long lists"
        self._md_hint = os.environ.get("CM3D_MD_HINT", "1") == "1"
        self._md_fb = torch.ones(4, dtype=torch.int32).pin_memory()
        self._md_fb_np = self._md_fb.numpy()
        self.prior_wlh = torch.from_numpy(self.classes.prior_wlh).to(d)
        self.is_vehicle = torch.from_numpy(self.classes.is_vehicle).to(d)
        self.nms_thr = torch.from_numpy(self.classes.nms_thr).to(d)
        self.nms_group = torch.from_numpy(np.ascontiguousarray(self.classes.nms_group, np.int32)).to(d)

    # -- upload + allocation
    def upload(self, hb: HostBatch, dense_masks: Optional[torch.Tensor] = None):
        d = self.dev
        def t(a):
            h = torch.from_numpy(np.ascontiguousarray(a))
            return h.to(d, non_blocking=h.is_pinned())        # page-locked staging buffers (cm3d_amd.reader) copy asynchronously
        F, M, S = hb.n_frames, hb.n_masks, len(hb.sweep_row_off) - 1
        if M <= 0 or F <= 0 or S <= 0 or hb.n_raw_rows <= 0:
            raise ValueError("empty batch")
        nm_max = int(np.diff(hb.mask_off).max())
        if hb.n_masks * hb.height * ((hb.width + 31) // 32) > 0x7FFFFFFF:
            raise ValueError(f"{hb.n_masks} masks of {hb.width}x{hb.height} in one batch: their bit-packed words no longer fit a 32-bit offset "
                             "(split the batch)")
        if nm_max > _lib.MAX_MASKS_PER_FRAME:
            raise ValueError(f"{nm_max} masks in one frame (limit {_lib.MAX_MASKS_PER_FRAME})")
        if hb.n_cams > _lib.MAX_CAMS or hb.mask_cam.min() < 0 or hb.mask_cam.max() >= hb.n_cams:
            raise ValueError("cam_nums out of range")
        W, H = hb.width, hb.height
        Wp = (W + 31) // 32
        b = type("DeviceBatch", (), {})()
        b.hb, b.F, b.M, b.S, b.W, b.H, b.Wp = hb, F, M, S, W, H, Wp
        b.sweep_row_off = t(hb.sweep_row_off); b.sweep_xf = t(hb.sweep_xf)
        b.frame_sweep_off = t(hb.frame_sweep_off)
        b.cams = t(hb.cams); b.mask_off = t(hb.mask_off); b.mask_cam = t(hb.mask_cam); b.mask_frame = t(hb.mask_frame)
        b.rle_off = t(hb.rle_off)
        b.class_id = t(hb.class_id); b.score = t(hb.score)
        b.pose_rt = t(hb.pose_rt) if hb.pose_rt is not None else None
        b.pose_inv = t(hb.pose_inv) if hb.pose_inv is not None else None
        b.halfw = self.halfw if hb.ego_box else 0.0
        b.frame_lane = t(hb.frame_lane); b.ego_xyz = t(hb.ego_xyz)
        b.pt_cap = hb.n_raw_rows
        b.max_pts = int(max(hb.sweep_row_off[hb.frame_sweep_off[1:]] - hb.sweep_row_off[hb.frame_sweep_off[:-1]]))
        b.planes = (nm_max + 31) // 32
        b.idx_cap = int(max(1024, self.hits_per_point * b.pt_cap))
        e = lambda *shape, dtype=torch.int32: torch.empty(*shape, dtype=dtype, device=d)
        # the transformed cloud exists only on request, or when the sweeps cannot be fused into the projection launch
        b.max_sweeps = int(np.max(np.diff(hb.frame_sweep_off))) if hb.frame_sweep_off.size > 1 else 0
        b.fused = self.fused_sweeps and 0 < b.max_sweeps <= _lib.MAX_FUSED_SWEEPS
        b.points = e(b.pt_cap, 4, dtype=torch.float32) if (self.keep_cloud or not b.fused) else None
        b.pt_off = e(F + 1)
        b.status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=d)
        b.packed = e(M, H, Wp)
        b.bbox = e(M, _lib.BBOX_STRIDE)        # [0..3] bounds of the eroded pixels, [4..7] the rectangle `packed` stores
        b.hit_words = e(b.planes, b.pt_cap)
        b.hit_count = e(M); b.hit_off = e(M + 1); b.tile_off = e(M + 1)
        b.hit_idx = e(b.idx_cap); b.hit_xyz = e(b.idx_cap, 4, dtype=torch.float32)
        b.removed_words = int(self.lib.cm3d_removed_words(b.pt_cap, F))
        b.removed_bits = e(b.removed_words)
        b.medoid_pos = e(M); b.centroid = e(M, 3, dtype=torch.float32)
        b.centroid_g = e(M, 3, dtype=torch.float32) if hb.pose_rt is not None else b.centroid
        b.colsum = e(b.idx_cap, dtype=torch.float32) if self.keep_colsum else None
        b.lane_idx = e(M); b.lane_dist = e(M, dtype=torch.float64)
        b.box = e(M, _lib.BOX_STRIDE, dtype=torch.float64); b.flags = e(M)
        L = self.lib
        b.tile_work = torch.empty(int(L.cm3d_tile_work_bytes(M, b.idx_cap)), dtype=torch.uint8, device=d)
        ws = max(L.cm3d_rle_workspace_bytes(max(1, hb.rle_counts.size)),
                 L.cm3d_medoid_workspace_bytes(M, b.idx_cap),
                 L.cm3d_lane_nn_workspace_bytes(M))
        b.ws_bytes = int(ws)
        b.ws = torch.empty(b.ws_bytes, dtype=torch.uint8, device=d)
        # per-(block, mask) hit counts: written by project_hits, consumed by compact_hits
        b.pg_ws_bytes = int(L.cm3d_project_workspace_bytes(F, b.max_pts, b.planes))
        b.pg_ws = torch.empty(max(b.pg_ws_bytes, 16), dtype=torch.uint8, device=d)
        # the RLE run ends stay alive across the whole pass, so they get their own buffer
        b.rle_ws_bytes = int(L.cm3d_rle_workspace_bytes(max(1, hb.rle_counts.size)))
        b.rle_ws = torch.empty(b.rle_ws_bytes, dtype=torch.uint8, device=d)
        b.n_tables, b.n_lane = len(hb.lane_off) - 1, int(hb.lane.shape[0])
        b.grid_bytes = int(L.cm3d_lane_grid_bytes(b.n_tables, b.n_lane))
        # The spatial index of the lane tables depends on the tables alone (the reference discretises a scene's lanes once,
        # 2d_to_3d.py:406, and looks every frame of the scene up in them): it is built once per distinct set of tables and kept
        # across passes and uploads -- consecutive batches of a scene, and every pass over a resident batch, reuse it.
        # (a caller that knows which tables these are says so -- pipeline_nuscenes: the map locations --; else their checksum)
        # (the key first: on a hit the tables are neither uploaded -- a pageable, host-blocking copy of megabytes -- nor read for a
        # checksum; the checksum of a table set is kept on the host batch.  The cache is shared by the engines of a LiftPipeline:
        # consecutive batches of a job alternate between them.)
        key = getattr(hb, "lane_key", None)
        if key is None:
            key = getattr(hb, "_lane_crc", None)
            if key is None:
                key = hb._lane_crc = (hb.lane.shape, hb.lane_off.tobytes(), zlib.crc32(np.ascontiguousarray(hb.lane).view(np.uint8)))
        hit = self._lane_cache.get(key)
        if hit is not None:
            self._lane = hit
            b.lane, b.lane_off, b.grid = hit["lane"], hit["lane_off"], hit["grid"]
        else:
            b.lane = t(hb.lane); b.lane_off = t(hb.lane_off)
            b.grid = torch.empty(b.grid_bytes, dtype=torch.uint8, device=d)
            up = torch.cuda.Event()
            up.record(torch.cuda.current_stream(d))         # whichever engine builds the index waits for the tables' copies (its streams are not this one)
            self._lane = {"key": key, "lane": b.lane, "lane_off": b.lane_off, "grid": b.grid, "built": False, "tables_uploaded": up,
                          "built_event": torch.cuda.Event()}
            if len(self._lane_cache) >= 8:                  # a handful of cities per job: keep the cache small
                self._lane_cache.pop(next(iter(self._lane_cache)))
            self._lane_cache[key] = self._lane
        b.dense = dense_masks
        # The bulk data LAST: sweeps and run lengths come from page-locked staging buffers (cm3d_amd.reader) and copy
        # asynchronously; the small arrays above are pageable, their copies block the host until everything queued before them on
        # the stream is done -- behind the 180 MB of a C2 batch's sweeps that was the whole transfer time (4 ms), per batch.
        b.rle_counts = t(hb.rle_counts.view(np.int32))
        b.intensity = t(hb.intensity) if hb.intensity is not None else None
        b.raw = t(hb.raw)
        self.b = b
        return b

    def decode_masks_dense(self):
        """a1: RLE -> dense uint8 (M,H,W) on the device (pycocotools.mask.decode, reference :425)."""
        b = self.b
        if b.dense is None:
            b.dense = torch.empty(b.M, b.H, b.W, dtype=torch.uint8, device=self.dev)
        st = torch.cuda.current_stream(self.dev).cuda_stream
        check(self.lib.cm3d_rle_to_dense(_ptr(b.rle_counts), _ptr(b.rle_off), b.M, b.hb.rle_counts.size, b.W, b.H,
                                         _ptr(b.dense), _ptr(b.rle_ws), b.rle_ws_bytes, st), "cm3d_rle_to_dense")
        return b.dense

    # -- the stages, in reference order
    def stage_begin(self, st, defer_reset=False):
        """Resets the per-pass state and starts the lane-grid build on the side stream.  defer_reset: the reset rides on the mask stage's
        launch instead (stage_masks(..., reset=True): run() does that for resident run lengths on the fused-sweeps path, one launch less
        per pass)."""
        b = self.b
        if not defer_reset:
            check(self.lib.cm3d_batch_begin(_ptr(b.status), _ptr(b.hit_count), b.M, _ptr(b.removed_bits), b.removed_words, st),
                  "cm3d_batch_begin")
        if self._lane["built"]:
            return                           # same lane tables as the last build: the index is still valid
        main = torch.cuda.current_stream(self.dev)
        self.side.wait_stream(main)          # the uploads of the tables have been issued before this point
        self.side.wait_event(self._lane["tables_uploaded"])     # (by another engine of the pipeline, on its stream, when the entry is shared)
        with torch.cuda.stream(self.side):
            self.stage_lane_grid(self.side.cuda_stream)
            self._lane["built_event"].record(self.side)
        self._lane["built"], self._lane["done"], self._lane["passes_since_build"] = True, False, 0

    def wait_lane_grid(self):
        """Orders the current stream behind the build of the lane index (by whichever engine of the pipeline built it; nothing to do
        once the host has seen a pass that followed the build complete: check_status / download / capture_graph note that)."""
        if not self._lane.get("done"):
            torch.cuda.current_stream(self.dev).wait_event(self._lane["built_event"])

    def rebuild_lane_grid(self):
        """Forgets the cached lane index: the next pass builds it again (benchmarks that want the build inside a pass)."""
        if self._lane is not None:
            self._lane["built"], self._lane["done"] = False, False

    def stage_sweeps(self, st):
        b = self.b
        check(self.lib.cm3d_sweep_prep(_ptr(b.raw), b.hb.raw_stride, _ptr(b.intensity), _ptr(b.sweep_row_off), b.S, b.hb.max_rows_per_sweep,
                                       _ptr(b.sweep_xf), _ptr(b.frame_sweep_off), b.F, b.halfw, _ptr(b.points), b.pt_cap,
                                       _ptr(b.pt_off), _ptr(b.removed_bits), _ptr(b.status), st), "cm3d_sweep_prep")

    def stage_masks(self, st, masks="dense", reset=False):
        """reset (run lengths only): this launch also does cm3d_batch_begin's work -- it must then be the first call of the pass."""
        b = self.b
        if reset:
            if masks != "rle":
                raise ValueError("the per-pass reset rides on the run-length mask launch only")
            check(self.lib.cm3d_rle_erode_pack_begin(_ptr(b.rle_counts), _ptr(b.rle_off), b.M, b.hb.rle_counts.size, b.W, b.H,
                                                     _ptr(b.packed), _ptr(b.bbox), _ptr(b.rle_ws), b.rle_ws_bytes,
                                                     _ptr(b.status), _ptr(b.hit_count), b.M, _ptr(b.removed_bits), b.removed_words, st),
                  "cm3d_rle_erode_pack_begin")
            return
        if masks == "dense":
            if b.dense is None:
                raise Cm3dError("dense masks requested but not resident: call decode_masks_dense() or pass dense_masks")
            check(self.lib.cm3d_erode_pack(_ptr(b.dense), b.M, b.W, b.H, _ptr(b.packed), _ptr(b.bbox), st), "cm3d_erode_pack")
        elif masks == "rle":
            check(self.lib.cm3d_rle_erode_pack(_ptr(b.rle_counts), _ptr(b.rle_off), b.M, b.hb.rle_counts.size, b.W, b.H,
                                               _ptr(b.packed), _ptr(b.bbox), _ptr(b.rle_ws), b.rle_ws_bytes, st),
                  "cm3d_rle_erode_pack")
        else:
            raise ValueError(masks)

    def stage_project(self, st, events=None):
        """events: optional pair of torch.cuda.Event (enable_timing) that the library records around the projection
        kernel itself on the launch stream (bench.py's roofline timing)."""
        b = self.b
        e0, e1 = self._raw_events(events)
        check(self.lib.cm3d_project_hits(_ptr(b.points), _ptr(b.pt_off), b.F, b.max_pts, b.pt_cap, _ptr(b.cams), b.hb.n_cams,
                                         _ptr(b.mask_off), _ptr(b.mask_cam), _ptr(b.bbox), _ptr(b.packed), b.M, b.W, b.H,
                                         self.min_dist, b.planes, _ptr(b.hit_words), _ptr(b.hit_count), _ptr(b.status),
                                         _ptr(b.pg_ws), b.pg_ws_bytes, e0, e1, st), "cm3d_project_hits")

    @staticmethod
    def _raw_events(events):
        """torch events -> the hipEvent_t handles the C-ABI takes (created by a first record on the current stream)."""
        if events is None:
            return 0, 0
        out = []
        for ev in events:
            if ev.cuda_event == 0 or ev.cuda_event is None:
                ev.record()
            out.append(int(ev.cuda_event))
        return out[0], out[1]

    def can_fuse_sweeps(self):
        return self.b.fused

    def stage_sweep_project(self, st, events=None):
        """Sweep preparation folded into the projection kernel (cm3d_sweep_project_hits): same outputs as
        stage_sweeps + stage_project, the cloud crosses HBM once less.  Needs the masks of the batch (stage_masks) first."""
        b = self.b
        e0, e1 = self._raw_events(events)
        check(self.lib.cm3d_sweep_project_hits(_ptr(b.raw), b.hb.raw_stride, _ptr(b.intensity), _ptr(b.sweep_row_off), b.S, b.max_sweeps, _ptr(b.sweep_xf),
                                               _ptr(b.frame_sweep_off), b.halfw, _ptr(b.points), b.pt_cap, _ptr(b.pt_off),
                                               _ptr(b.removed_bits), b.F, b.max_pts, b.pt_cap, _ptr(b.cams),
                                               b.hb.n_cams, _ptr(b.mask_off), _ptr(b.mask_cam), _ptr(b.bbox), _ptr(b.packed), b.M, b.W,
                                               b.H, self.min_dist, b.planes, _ptr(b.hit_words), _ptr(b.hit_count), _ptr(b.status),
                                               _ptr(b.pg_ws), b.pg_ws_bytes, e0, e1, st), "cm3d_sweep_project_hits")

    def stage_compact(self, st):
        b = self.b
        # coordinates of the in-mask points: from the cloud when it exists, else re-derived from the raw rows
        from_raw = b.points is None
        check(self.lib.cm3d_compact_hits(_ptr(b.hit_words), b.planes, b.F, b.max_pts, b.pt_cap, _ptr(b.mask_off), b.M, _ptr(b.hit_count),
                                         _ptr(b.removed_bits), _ptr(b.raw) if from_raw else 0, b.hb.raw_stride,
                                         _ptr(b.intensity) if from_raw else 0, _ptr(b.sweep_xf) if from_raw else 0, _ptr(b.points), _ptr(b.hit_off), _ptr(b.tile_off),
                                         _ptr(b.hit_idx), 0, _ptr(b.hit_xyz), b.idx_cap, _ptr(b.tile_work), _ptr(b.status),
                                         _ptr(b.pg_ws), b.pg_ws_bytes, st), "cm3d_compact_hits")

    def stage_medoid(self, st):
        """The medoid stage.  In a batch with a list of more than 448 points, lists of more than 256 go a two-pass route that costs two
        launches even when a batch holds no such list (they find that out on the device: 2-3 % of a pass with three batches in flight).  The first launch leaves word 0
        of `self._md_fb` -- page-locked host memory the device writes directly, no copy -- saying whether THIS batch held one; the next
        pass of this engine reads whatever has arrived by then and, if the last batch it heard of had none, asks for the one-pass
        route only (cm3d_medoid2 flags bit 0).  That route is exact for every length: a stale hint costs time on one pass, never a
        result."""
        b = self.b
        flags = 1 if (self._md_hint and int(self._md_fb_np[0]) == 0) else 0
        check(self.lib.cm3d_medoid2(_ptr(b.hit_xyz), 0, 0, b.M, _ptr(b.hit_off), _ptr(b.tile_off), 0, b.idx_cap, _ptr(b.tile_work),
                                    _ptr(b.medoid_pos), _ptr(b.centroid), _ptr(b.colsum), _ptr(b.ws), b.ws_bytes, flags,
                                    self._md_fb.data_ptr(), st), "cm3d_medoid2")

    def stage_lane_grid(self, st):
        """Spatial index of the lane tables.  It depends on the lane tables only, so `run` issues it on
        the side stream at the start of the pass; it overlaps the sweep / mask / projection stages."""
        b = self.b
        check(self.lib.cm3d_lane_grid_build(_ptr(b.lane), _ptr(b.lane_off), b.n_tables, b.n_lane, _ptr(b.grid), b.grid_bytes, st),
              "cm3d_lane_grid_build")

    def stage_lanes(self, st):
        b = self.b
        if b.pose_rt is not None:       # Waymo: the lane lookup happens in the global frame
            check(self.lib.cm3d_centroid_transform(_ptr(b.centroid), _ptr(b.medoid_pos), _ptr(b.mask_frame), b.M, _ptr(b.pose_rt),
                                                   _ptr(b.centroid_g), st), "cm3d_centroid_transform")
        check(self.lib.cm3d_lane_nn(_ptr(b.centroid_g), _ptr(b.medoid_pos), _ptr(b.mask_frame), b.M, _ptr(b.lane), _ptr(b.lane_off),
                                    _ptr(b.frame_lane), b.n_tables, b.n_lane, _ptr(b.grid), _ptr(b.lane_idx),
                                    _ptr(b.lane_dist), _ptr(b.ws), b.ws_bytes, st), "cm3d_lane_nn")

    def stage_boxes(self, st):
        b = self.b
        check(self.lib.cm3d_box_nms(_ptr(b.centroid_g), _ptr(b.medoid_pos), _ptr(b.mask_off), b.F, b.M, _ptr(b.class_id),
                                    _ptr(b.score), _ptr(b.lane), _ptr(b.lane_off), _ptr(b.frame_lane), _ptr(b.lane_idx),
                                    _ptr(b.lane_dist), _ptr(self.prior_wlh), _ptr(self.is_vehicle), _ptr(self.nms_group),
                                    _ptr(self.nms_thr), len(self.classes.names), _ptr(b.ego_xyz), _ptr(b.pose_inv), _ptr(b.box),
                                    _ptr(b.flags), st), "cm3d_box_nms")

    STAGES = ("sweeps", "masks", "project", "compact", "medoid", "lanes", "boxes")

    def run(self, masks="dense", project_events=None, stage_events=None):
        """One pass of the hot path over the resident batch (asynchronous).  project_events: optional pair of
        torch.cuda.Event recorded around the projection launch on the launch stream (bench.py's roofline timing).
        stage_events: optional list; gets five timing events appended -- before the point/mask stages, behind the compaction,
        the medoid, the lane search and the boxes -- the entry points' timer buckets (reference :368-378) come from them."""
        st = torch.cuda.current_stream(self.dev).cuda_stream

        def mark():
            if stage_events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                stage_events.append(e)
        # (resident run lengths on the fused-sweeps path: the mask launch is the pass's first and carries the reset -- one launch and one
        # launch boundary less per pass; CM3D_FUSED_RESET=0: the reset as a launch of its own)
        fused_reset = masks == "rle" and self.can_fuse_sweeps() and os.environ.get("CM3D_FUSED_RESET", "1") != "0"
        self.stage_begin(st, defer_reset=fused_reset)
        mark()
        if not self.can_fuse_sweeps():
            self.stage_sweeps(st)
        self.stage_masks(st, masks, reset=fused_reset)
        if not self.can_fuse_sweeps():
            self.stage_project(st, project_events)
        else:
            self.stage_sweep_project(st, project_events)
        self.stage_compact(st)
        mark()
        self.stage_medoid(st)
        mark()
        self.wait_lane_grid()
        self.stage_lanes(st)
        mark()
        self.stage_boxes(st)
        mark()
        self._lane["passes_since_build"] = self._lane.get("passes_since_build", 0) + 1

    def capture_graph(self, masks="rle"):
        """Captures one pass over the resident batch into a HIP graph and returns it (`g.replay()` re-runs the pass
        on the resident buffers).  One graph launch replaces ~13 kernel launches, two event operations and the Python
        between them: what matters for small batches, where a pass is shorter than the time the host needs to enqueue
        it (a 256-frame batch is GPU-bound either way).  Call after at least one eager `run` (first-use attribute
        calls and stream creation must not happen during capture)."""
        torch.cuda.synchronize(self.dev)
        if self._lane is not None and self._lane["built"]:
            self._lane["done"] = True            # everything issued so far has completed, the lane index among it
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run(masks=masks)
        return g

    def hit_chunk_rows(self) -> int:
        """Rows of the resident batch that lie in a 256-row block with at least one in-mask point after the last pass: the rows
        whose hit words the projection wrote and the compaction read (bench.py's byte accounting; synchronises the stream)."""
        import ctypes
        b = self.b
        out = ctypes.c_int64(0)
        check(self.lib.cm3d_project_hit_rows(_ptr(b.pg_ws), b.pg_ws_bytes, b.F, b.max_pts, b.planes, ctypes.addressof(out),
                                             torch.cuda.current_stream(self.dev).cuda_stream), "cm3d_project_hit_rows")
        return int(out.value)

    # -- results
    def check_status(self):
        s = self.b.status.cpu().numpy()
        if self._lane is not None and self._lane["built"] and self._lane.get("passes_since_build", 0) > 0:
            self._lane["done"] = True        # a pass that waited for the build has completed
        if s[0] & 1:
            raise Cm3dError(f"point capacity overflow ({s[1]} > {self.b.pt_cap})")
        if s[0] & 2:
            raise Cm3dError(f"hit-index capacity overflow: {s[2]} indices needed, capacity {self.b.idx_cap}; "
                            "raise hits_per_point")
        if s[0] & 4:
            raise Cm3dError("a frame has too many masks or a cam_num is out of range")
        if s[0] & 8:
            raise Cm3dError("quad layout: a frame does not start on a multiple of 4 rows")
        return s

    def removed_rows(self):
        """Boolean array over the batch's raw rows: True where the reference drops the row (ego box, :442-445).
        Decoded from the device's removed-row bits (frame f's bits start at word (pt_off[f] >> 5) + 8 f)."""
        b = self.b
        words = b.removed_bits.cpu().numpy().view(np.uint32)
        p_off = b.pt_off.cpu().numpy()
        out = np.zeros(int(p_off[-1]), bool)
        for f in range(b.F):
            p0, n = int(p_off[f]), int(p_off[f + 1] - p_off[f])
            if n <= 0:
                continue
            w0 = (p0 >> 5) + 8 * f
            bits = np.unpackbits(words[w0:w0 + (n + 31) // 32].view(np.uint8), bitorder="little")[:n]
            out[p0:p0 + n] = bits.astype(bool)
        return out

    def download(self, full=True):
        """Synchronises, checks the status word and returns numpy results.  full=False: only the per-mask results
        (boxes, flags, medoids, lane matches, list offsets) -- a few hundred KB.  full=True adds the index lists
        (`hit_idx`, the reference's track_points), the coordinates of the listed points (`hit_xyz`), `pt_off` in the
        reference's form (the aggregated cloud without the ego-box rows, :445-465) and -- when the engine keeps the cloud
        (keep_cloud) -- `points` in that form too; on the device the cloud keeps the dropped rows as NaN placeholders."""
        b = self.b
        s = self.check_status()
        if not full:
            return dict(hit_off=b.hit_off.cpu().numpy(), medoid_pos=b.medoid_pos.cpu().numpy(), centroid=b.centroid.cpu().numpy(),
                        lane_idx=b.lane_idx.cpu().numpy(), lane_dist=b.lane_dist.cpu().numpy(),
                        centroid_global=b.centroid_g.cpu().numpy(), box=b.box.cpu().numpy(), flags=b.flags.cpu().numpy())
        n_rows, n_idx = int(s[1]), int(s[2])
        pt_off_rows = b.pt_off.cpu().numpy()
        keep = ~self.removed_rows()[:n_rows]
        if b.hb.frame_rows is not None:         # quad layout: a frame's padding rows are not rows of the reference's cloud
            for f in range(b.F):
                p0 = int(pt_off_rows[f])
                keep[p0 + int(b.hb.frame_rows[f]):int(pt_off_rows[f + 1])] = False
        kept_per_frame = np.add.reduceat(keep.astype(np.int64), pt_off_rows[:-1]) if n_rows else np.zeros(b.F, np.int64)
        kept_per_frame = np.where(np.diff(pt_off_rows) > 0, kept_per_frame, 0)
        out = dict(
            pt_off=np.concatenate([[0], np.cumsum(kept_per_frame)]).astype(np.int32),
            hit_off=b.hit_off.cpu().numpy(), hit_idx=b.hit_idx[:n_idx].cpu().numpy(), hit_xyz=b.hit_xyz[:n_idx].cpu().numpy(),
            bbox=b.bbox[:, :4].cpu().numpy(), medoid_pos=b.medoid_pos.cpu().numpy(), centroid=b.centroid.cpu().numpy(),
            lane_idx=b.lane_idx.cpu().numpy(), lane_dist=b.lane_dist.cpu().numpy(), centroid_global=b.centroid_g.cpu().numpy(),
            box=b.box.cpu().numpy(), flags=b.flags.cpu().numpy())
        if b.points is not None:
            out["points"] = b.points[:n_rows].cpu().numpy()[keep]
        if b.colsum is not None:
            out["colsum"] = b.colsum[:n_idx].cpu().numpy()
        return out


class LiftPipeline:
    """Keeps `depth` independent lift batches in flight: one LiftEngine (own device buffers) per HIP stream, batches
    issued round-robin.  Frames are independent, so consecutive batches of a job never wait for each other;
    a pass is a chain of ~12 dependent launches whose tail (scans, lane search, boxes: a few hundred waves) leaves most
    of the 256 CUs idle, and the next batch's HBM-bound stages run there.  Measured on the C2 batch: 863 k frames/s with
    one batch in flight, 997 k with two, 1.05-1.08 M with three (four or more: no further gain)."""

    def __init__(self, device="cuda:0", depth=3, **engine_kw):
        if depth < 1:
            raise ValueError("depth >= 1")
        self.dev = torch.device(device)
        shared = {}                                          # one cache of lane tables + indices for all slots
        self.engines = [LiftEngine(device, lane_cache=shared, **engine_kw) for _ in range(depth)]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(depth)]
        self.masks = [None] * depth
        self.uploaded = [torch.cuda.Event() for _ in range(depth)]      # recorded behind a slot's H2D copies (submit)
        self._next = 0
        # With batches in flight the projection launch leaves part of the chip to the other batches' kernels (two workgroups per CU
        # instead of three, one with four batches in flight: cm3d_project_workgroups_per_cu -- process-wide, results unaffected): +1-2 %
        # frames/s at depth 3, +4 % at depth 4, while one batch at a time runs fastest with the launch filling the chip.
        # CM3D_PIPE_WG_PER_CU overrides (0: never ask).  Four batches in flight need GPU_MAX_HW_QUEUES >= 8 in the environment before the HIP
        # runtime starts (bench.py sets it): on the default four hardware queues a fourth stream shares one and the pass gets slower.
        want = int(os.environ.get("CM3D_PIPE_WG_PER_CU", "1" if depth >= 4 else "2"))       # (four in flight: 2.51 M frames/s at 1, 2.46 at 2, 2.40 at 3)
        if depth >= 2 and want > 0:
            self.engines[0].lib.cm3d_project_workgroups_per_cu(want)
        elif depth == 1:
            self.engines[0].lib.cm3d_project_workgroups_per_cu(0)

    @property
    def depth(self):
        return len(self.engines)

    def submit(self, hb: HostBatch, masks="rle", stage_events=None):
        """Uploads `hb` into the next slot and issues one pass over it on that slot's stream (asynchronous).
        Returns the slot; `collect(slot)` must be called before the slot comes round again."""
        slot = self._next
        self._next = (slot + 1) % self.depth
        eng = self.engines[slot]
        with torch.cuda.stream(self.streams[slot]):
            eng.upload(hb)
            self.uploaded[slot].record(self.streams[slot])
            if masks == "dense":
                eng.decode_masks_dense()
            eng.run(masks=masks, stage_events=stage_events)
        self.masks[slot] = masks
        return slot

    def rerun(self, slot, masks=None, project_events=None):
        """Another pass over the batch resident in `slot` (benchmarks)."""
        with torch.cuda.stream(self.streams[slot]):
            self.engines[slot].run(masks=masks or self.masks[slot], project_events=project_events)

    def collect_records(self, slot, frame_ids):
        """Waits for the slot's stream, checks the status word and returns the slot's kept-box records as a DEVICE tensor
        (kept_box_records): what an entry point hands to the end-of-job gather -- nothing else is downloaded."""
        self.streams[slot].synchronize()
        eng = self.engines[slot]
        eng.check_status()
        with torch.cuda.stream(self.streams[slot]):
            rec = kept_box_records(eng.b, frame_ids, self.dev)
        self.streams[slot].synchronize()
        return rec

    def collect(self, slot, full=True):
        """Waits for the slot's stream only and returns (host batch, numpy results); full=False: per-mask results only
        (LiftEngine.download)."""
        self.streams[slot].synchronize()
        eng = self.engines[slot]
        with torch.cuda.stream(self.streams[slot]):
            res = eng.download(full=full)
        return eng.b.hb, res


REC_FRAME_A, REC_FRAME_B = 5, 6       # columns of a shipped record that carry the frame's identity (see kept_box_records)


def kept_box_records(b, frame_ids, device=None):
    """The fixed-size records the multi-GPU gather ships: one row of CM3D_BOX_STRIDE doubles per box that
    survives NMS, in (frame, mask) order.  Columns 0-4, 7-9 are cm3d_box_nms's (centre, rotation, score, class, flags);
    columns 5 and 6 -- lane yaw and distance, which no output file contains -- are overwritten with the frame's identity
    frame_ids[f] = (a, b) (nuScenes: global sample index, 0; Waymo: scene index, frame number), so that rank 0 can rebuild
    every output record from the gathered rows and the job's deterministic frame order alone."""
    device = device or b.box.device
    keep = (b.flags & 3) == 3
    rec = b.box[keep].clone()
    ids = torch.as_tensor(np.ascontiguousarray(frame_ids, np.float64).reshape(-1, 2), device=device)
    fr = b.mask_frame[keep].long()
    rec[:, REC_FRAME_A] = ids[fr, 0]
    rec[:, REC_FRAME_B] = ids[fr, 1]
    return rec


def nuscenes_boxes_from_records(rec, tokens, classes: Optional[ClassTable] = None):
    """Gathered records (numpy (k, 10), see kept_box_records) -> the reference's per-sample box dict lists (:808-817 after NMS
    :913-924): {sample_token: [box, ...]} with every token of `tokens` present ([] when a sample has no box, :845)."""
    classes = classes or ClassTable.nuscenes()
    results = {t: [] for t in tokens}
    # plain Python floats / ints in one go (tolist), per-class constants built once: the loop below only assembles dicts
    rows = np.asarray(rec, np.float64).reshape(-1, _lib.BOX_STRIDE).tolist()
    sizes = [[float(v) for v in wlh] for wlh in classes.prior_wlh]
    for r in rows:
        token, ci = tokens[int(r[REC_FRAME_A])], int(r[8])
        name = classes.names[ci]
        results[token].append({
            "sample_token": token,
            "translation": [r[0], r[1], r[2]],
            "size": list(sizes[ci]),
            "rotation": [r[3], 0.0, 0.0, r[4]],
            "velocity": [0, 0],
            "detection_name": name,
            "detection_score": r[7],
            "attribute_name": ATTRIBUTE_NAMES[name],
        })
    return results


def nuscenes_results_json(rec, tokens, classes: Optional[ClassTable] = None, meta=None):
    """The text json.dumps({"meta": meta, "results": nuscenes_boxes_from_records(rec, tokens, classes)}) produces, written out
    directly from the records: same keys, key order, separators and float repr -- the writer of the reference (:929-930) without
    60 000 dicts and a generic encoder in between (tests/test_host_logic.py holds the two against each other)."""
    import json
    classes = classes or ClassTable.nuscenes()
    rows = np.asarray(rec, np.float64).reshape(-1, _lib.BOX_STRIDE).tolist()
    tok_js = [json.dumps(t) for t in tokens]
    # everything of a box that only depends on its class, rendered once
    tail = []
    for ci, name in enumerate(classes.names):
        size = json.dumps([float(v) for v in classes.prior_wlh[ci]])
        tail.append((size, f'"velocity": [0, 0], "detection_name": {json.dumps(name)}, "detection_score": ',
                     f', "attribute_name": {json.dumps(ATTRIBUTE_NAMES[name])}}}'))
    per = [[] for _ in tokens]
    from math import isfinite

    def fr(x):                      # json's float text: repr, but Infinity / -Infinity / NaN for what repr calls inf / nan
        return float.__repr__(x) if isfinite(x) else json.dumps(x)
    for r in rows:
        ti, ci = int(r[REC_FRAME_A]), int(r[8])
        size, mid, end = tail[ci]
        x, y, z, qw, qz, sc = r[0], r[1], r[2], r[3], r[4], r[7]
        if isfinite(x + y + z + qw + qz + sc):          # (a sum that overflows only sends a finite row down the careful branch)
            per[ti].append(f'{{"sample_token": {tok_js[ti]}, "translation": [{x!r}, {y!r}, {z!r}], "size": {size}, '
                           f'"rotation": [{qw!r}, 0.0, 0.0, {qz!r}], {mid}{sc!r}{end}')
        else:
            per[ti].append(f'{{"sample_token": {tok_js[ti]}, "translation": [{fr(x)}, {fr(y)}, {fr(z)}], "size": {size}, '
                           f'"rotation": [{fr(qw)}, 0.0, 0.0, {fr(qz)}], {mid}{fr(sc)}{end}')
    body = ", ".join(f'{tok_js[i]}: [{", ".join(b)}]' for i, b in enumerate(per))
    return f'{{"meta": {json.dumps(meta if meta is not None else {})}, "results": {{{body}}}}}', sum(len(b) for b in per)


def nuscenes_results_json_native(rec, tokens, classes: Optional[ClassTable] = None, meta=None, part=None) -> bytes:
    """nuscenes_results_json through the native writer (libcm3d_reader.so, cm3d_write_results_json): the same bytes
    (tests/test_reader.py), without a Python statement per box -- 60 000 boxes take milliseconds instead of a third of a second.
    part = (first, count): only the entries of tokens[first:first+count] (records whose column 5 lies in that range), without the
    file's head and tail -- the entry point formats every batch's share while the later batches are still on the GPU and joins
    the parts with ", " between nuscenes_results_json_head(meta) and b"}}"."""
    import json
    from . import reader as rdmod
    classes = classes or ClassTable.nuscenes()
    mid, score, tail = [], [], []
    for ci, name in enumerate(classes.names):
        size = json.dumps([float(v) for v in classes.prior_wlh[ci]])
        mid.append(f'], "size": {size}, "rotation": [')
        score.append(f'], "velocity": [0, 0], "detection_name": {json.dumps(name)}, "detection_score": ')
        tail.append(f', "attribute_name": {json.dumps(ATTRIBUTE_NAMES[name])}}}')
    if part is not None:
        first, count = part
        rec = np.array(rec, np.float64).reshape(-1, _lib.BOX_STRIDE)
        rec[:, REC_FRAME_A] -= first
        return rdmod.write_results_json(rec, [json.dumps(t) for t in tokens[first:first + count]], mid, score, tail, None)
    return rdmod.write_results_json(rec, [json.dumps(t) for t in tokens], mid, score, tail, nuscenes_results_json_head(meta).decode())


def nuscenes_results_json_head(meta=None) -> bytes:
    import json
    return f'{{"meta": {json.dumps(meta if meta is not None else {})}, "results": {{'.encode()


def box_records(hb: HostBatch, res: dict, classes: Optional[ClassTable] = None):
    """Device results -> the reference's per-sample box dict lists (:808-817, after NMS :913-924).
    Every sample keeps its key; a sample without boxes maps to [] (:845 runs before the `continue` at :896)."""
    classes = classes or ClassTable.nuscenes()
    results = {}
    for f, token in enumerate(hb.tokens):
        boxes = []
        for m in range(hb.mask_off[f], hb.mask_off[f + 1]):
            if (res["flags"][m] & 3) != 3:
                continue
            name = classes.names[hb.class_id[m]]
            bx = res["box"][m]
            boxes.append({
                "sample_token": token,
                "translation": [float(bx[0]), float(bx[1]), float(bx[2])],
                "size": [float(v) for v in classes.prior_wlh[hb.class_id[m]]],
                "rotation": [float(bx[3]), 0.0, 0.0, float(bx[4])],
                "velocity": [0, 0],
                "detection_name": name,
                "detection_score": float(hb.score[m]),
                "attribute_name": ATTRIBUTE_NAMES[name],
            })
        results[token] = boxes
    return results
