"""Waymo-side host pieces of the lifting path (reference src/waymo/2d_to_3d.py).

Only the deltas against nuScenes live here: the single-stage camera record (:561-593), lane yaws
from map polylines (:374-388), the frame-pose records used to move medoids to the global frame and
back (:684-690, :812-816), the class map (cfg/prompt_cfg.py:286-297), the per-type NMS thresholds
(:1147-1158) and the `metrics_pb2.Objects` writer (:1034-1065, :1300-1305).

Reading TFRecords needs waymo_open_dataset + TensorFlow (third-party, stays on the host, not in this
image); `frame_from_extracted` takes the same quantities already extracted per frame.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import geometry as geo

NUSC_TO_WAYMO = {   # cfg/prompt_cfg.py:286-297
    "car": "vehicle", "truck": "vehicle", "bus": "vehicle", "bicycle": "cyclist", "pedestrian": "pedestrian",
    "trailer": "vehicle", "barrier": "", "construction_vehicle": "vehicle", "traffic_cone": "", "motorcycle": "vehicle",
}
# label_pb2.Label.Type
WAYMO_TYPE = {"unknown": 0, "vehicle": 1, "pedestrian": 2, "sign": 3, "cyclist": 4}
THRESHS_BY_TYPE = {0: 1, 3: 0.175, 4: 0.85, 2: 0.175, 1: 4}     # 2d_to_3d.py:1147-1158 (squared metres)
RATIO = 1024 / 1920                                            # :523


def _quat_roundtrip_rotation(R32):
    """:569-575 / :686-689: scipy from_matrix -> as_quat -> pyquaternion(w,x,y,z).rotation_matrix.
    Returns float64 3x3 (the caller casts to float32)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(np.asarray(R32, np.float64)).as_quat()      # x,y,z,w
    return geo.quat_to_rotmat([q[3], q[0], q[1], q[2]])


def cam_record(extrinsic_row_major16, intrinsic, ratio=RATIO):
    """Camera record of one Waymo camera (:561-593) with the reference's own torch float32 ops."""
    axes = torch.tensor([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    axes = torch.linalg.inv(axes)                                                              # :566
    T = torch.from_numpy(np.array(extrinsic_row_major16, np.float64).reshape(4, 4)).to(dtype=torch.float32)
    T = torch.matmul(T, axes)                                                                  # :568
    Rq = _quat_roundtrip_rotation(T[:3, :3].numpy())                                           # :569-571
    t_added = (-T[:3, 3]).numpy().astype(np.float32)                                           # :575
    R_used = torch.from_numpy(Rq.T.copy()).to(dtype=torch.float32).numpy()                     # :576
    m = np.array(intrinsic, np.float32).tolist()                                               # :586
    K = np.array([[m[0], 0, m[2]], [0, m[1], m[3]], [0, 0, 1]]) * ratio                        # :587-588 (float64)
    K[2, 2] = 1
    K32 = torch.from_numpy(K).to(dtype=torch.float32).numpy()                                  # :593
    return geo.single_stage_cam_record(t_added, R_used, K32)


def get_yaws_from_lane_coords(polyline_xyz):
    """:374-388: finite-difference yaw along a lane polyline; returns (n,3) x,y,yaw."""
    prev_x, prev_y = 0, 0
    out = []
    for x, y, *_ in polyline_xyz:
        out.append([x, y, np.arctan2(y - prev_y, x - prev_x)])
        prev_x, prev_y = x, y
    if len(out) > 1:
        out[0][2] = out[1][2]
    return np.array(out, np.float64).reshape(-1, 3)


def pose_records(pose16):
    """frame.pose.transform -> (pose_rt float32[12], pose_inv float32[16]).
    pose_rt: the float32 rotation the reference passes to `rotate` (quaternion round trip, :686-689)
    and the translation it passes to `translate` (:690); pose_inv: np.linalg.inv of the float32 pose (:813)."""
    P = np.array(pose16, np.float32).reshape(4, 4)
    R = _quat_roundtrip_rotation(P[:3, :3]).astype(np.float32)
    rt = np.concatenate([R.reshape(9), P[:3, 3]]).astype(np.float32)
    inv = np.linalg.inv(P).astype(np.float32).reshape(16)
    return rt, inv


def frame_from_extracted(token, points_xyz, cams, rles, labels, scores, cam_nums, pose16, width, height, timestamp_micros=0,
                         context_name=""):
    """Kernel inputs of one Waymo frame.  points_xyz: (N,3) vehicle-frame TOP-lidar first returns
    (:472-479); the 4th column is ones like the reference's hstack (:477).  cams: list of
    (extrinsic16, intrinsic) in camera-name order 1..5 (:513-518)."""
    pts = np.concatenate([np.asarray(points_xyz, np.float32).reshape(-1, 3), np.ones((len(points_xyz), 1), np.float32)], 1)
    xf = np.zeros((1, geo.SWEEP_XF_STRIDE), np.float32)
    xf[0, 0:9] = np.eye(3, dtype=np.float32).reshape(9)
    xf[0, 12:21] = np.eye(3, dtype=np.float32).reshape(9)
    return SimpleNamespace(token=token, sweeps_raw=[pts], sweep_xf=xf,
                           cams=np.stack([cam_record(e, i) for e, i in cams]), rles=list(rles), labels=list(labels),
                           scores=[float(np.float32(s)) for s in scores],      # o.score is a proto float (:1049)
                           cam_nums=list(cam_nums), ego_xyz=np.zeros(3), width=int(width), height=int(height),
                           pose=np.array(pose16, np.float64).reshape(16), timestamp_micros=int(timestamp_micros),
                           context_name=context_name)


# ---------------------------------------------------------------- TFRecord route (third-party devkit, when installed)
def devkit_available():
    """waymo_open_dataset + TensorFlow importable?  (Neither is part of this repository's image; where they are installed the
    Waymo entry point reads the TFRecords itself, like the reference.)"""
    try:
        import tensorflow.compat.v1  # noqa: F401
        from waymo_open_dataset import dataset_pb2  # noqa: F401
        from waymo_open_dataset.utils import frame_utils  # noqa: F401
        return True
    except Exception:
        return False


def frame_records_from_tfrecord(path, frame_filter=None):
    """Yields (frame_num, record) for the frames of one Waymo TFRecord -- exactly the extraction steps of the reference's
    src/waymo/2d_to_3d.py:444-446 (Frame proto), :472-479 (TOP-lidar first returns through the devkit's range-image
    conversion), :513-518 (camera calibrations by name), :459-468 (lane polylines, frame 0 only) and nothing else.  A record
    holds what `<f>_frame.npz` holds (tools/extract_waymo_frames.py writes these very dicts to disk):
        points (N,3) f32, extrinsics (5,16), intrinsics (5,9), pose (16,), timestamp_micros, context_name [, lanes, lane_off]
    frame_filter(frame_num) -> False: the frame is still yielded (calibrations, pose, timestamp) but with an empty cloud -- the
    expensive range-image conversion of a frame nobody lifts (no mask files) is skipped."""
    import tensorflow.compat.v1 as tf
    from waymo_open_dataset import dataset_pb2
    from waymo_open_dataset.utils import frame_utils
    tf.enable_eager_execution()
    for fnum, data in enumerate(tf.data.TFRecordDataset(path, compression_type="")):
        frame = dataset_pb2.Frame()
        frame.ParseFromString(bytearray(data.numpy()))
        rec = {}
        if fnum == 0:
            polys = [np.array([[p.x, p.y, p.z] for p in f.lane.polyline]) for f in frame.map_features if f.HasField("lane")]
            rec["lanes"] = np.vstack(polys) if polys else np.zeros((0, 3))
            rec["lane_off"] = np.concatenate([[0], np.cumsum([len(p) for p in polys])]).astype(np.int64)
        if frame_filter is None or frame_filter(fnum):
            ri, cp, _, top_pose = frame_utils.parse_range_image_and_camera_projection(frame)
            pts, _ = frame_utils.convert_range_image_to_point_cloud(frame, ri, cp, top_pose, 0, False)
            rec["points"] = np.asarray(pts[0], np.float32)
        else:
            rec["points"] = np.zeros((0, 3), np.float32)
        cals = sorted(frame.context.camera_calibrations, key=lambda c: c.name)
        rec.update(extrinsics=np.array([list(c.extrinsic.transform) for c in cals]), intrinsics=np.array([list(c.intrinsic) for c in cals]),
                   pose=np.array(frame.pose.transform), timestamp_micros=np.int64(frame.timestamp_micros),
                   context_name=np.str_(frame.context.name))
        yield fnum, rec


# ---------------------------------------------------------------- metrics_pb2.Objects writer
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _double(field, v):
    return _key(field, 1) + np.float64(v).tobytes()


def _float(field, v):
    return _key(field, 5) + np.float32(v).tobytes()


def encode_object(center, length, width, height, heading, type_id, score, context_name, timestamp_micros,
                  object_id="unique object tracking ID"):
    """One waymo_open_dataset.protos.metrics_pb2.Object, hand-encoded in protobuf wire format.
    Field numbers from the public waymo-open-dataset protos (label.proto: Box center_x=1, center_y=2,
    center_z=3, width=4, length=5, height=6, heading=7; Label box=1, type=3, id=4; metrics.proto:
    Object object=1, score=2, context_name=4, frame_timestamp_micros=5).  The protos are not in the
    reference checkout: parity unpinned."""
    box = (_double(1, center[0]) + _double(2, center[1]) + _double(3, center[2]) + _double(4, width) + _double(5, length) +
           _double(6, height) + _double(7, heading))
    label = _ld(1, box) + _key(3, 0) + _varint(type_id) + _ld(4, object_id.encode())
    return _ld(1, label) + _float(2, score) + _ld(4, context_name.encode()) + _key(5, 0) + _varint(timestamp_micros)


def encode_objects(objs):
    """metrics_pb2.Objects.SerializeToString() for a list of encoded Object payloads (:1300-1305)."""
    return b"".join(_ld(1, o) for o in objs)


def objects_from_results(hb, res, classes, frame_meta):
    """Device results -> list of encoded Objects in the reference's order (frames in order, kept boxes in
    mask order; :1034-1065, :1262-1297).  frame_meta: per frame (context_name, timestamp_micros)."""
    out = []
    for f in range(hb.n_frames):
        ctx, ts = frame_meta[f]
        for m in range(hb.mask_off[f], hb.mask_off[f + 1]):
            if (res["flags"][m] & 3) != 3:
                continue
            ci = hb.class_id[m]
            pr = classes.prior_wlh[ci]
            b = res["box"][m]
            out.append(encode_object(b[0:3], length=pr[1], width=pr[0], height=pr[2], heading=b[3],
                                     type_id=WAYMO_TYPE[classes.out_names[ci]], score=hb.score[m], context_name=ctx,
                                     timestamp_micros=ts))
    return out


def decode_objects(blob):
    """Inverse of encode_objects for the fields this writer emits (tests, debugging)."""
    def rd_varint(b, i):
        v, sh = 0, 0
        while True:
            c = b[i]; i += 1
            v |= (c & 0x7F) << sh; sh += 7
            if not c & 0x80:
                return v, i

    def fields(b):
        i, out = 0, []
        while i < len(b):
            key, i = rd_varint(b, i)
            f, w = key >> 3, key & 7
            if w == 0:
                v, i = rd_varint(b, i)
            elif w == 1:
                v = float(np.frombuffer(b[i:i + 8], np.float64)[0]); i += 8
            elif w == 5:
                v = float(np.frombuffer(b[i:i + 4], np.float32)[0]); i += 4
            elif w == 2:
                n, i = rd_varint(b, i)
                v = bytes(b[i:i + n]); i += n
            else:
                raise ValueError("unsupported wire type")
            out.append((f, v))
        return out

    objs = []
    for f, payload in fields(blob):
        o = dict(fields(payload))
        label = dict(fields(o[1]))
        box = dict(fields(label[1]))
        objs.append(dict(center=[box[1], box[2], box[3]], width=box[4], length=box[5], height=box[6], heading=box[7],
                         type=label[3], id=label[4].decode(), score=o[2], context_name=o[4].decode(), timestamp_micros=o[5]))
    return objs
