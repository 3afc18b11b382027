"""Host-side geometry helpers used to assemble kernel inputs.

These restate the small pieces of pyquaternion 0.9.9 / nuscenes-devkit that the
reference leans on at its boundary with the kernels (reference
src/nuscenes/2d_to_3d.py:451-457,570-577,585-587).  The kernels consume the
float32 matrices produced here, never quaternions.
"""
import math

import numpy as np

CAM_STRIDE = 64          # floats per camera record (include/cm3d_hip.h CM3D_CAM_STRIDE)
CAM_K = 45               # offset of the 3x3 intrinsics; [54] = number of stages, [55] = translation flags
SWEEP_XF_STRIDE = 24     # floats per sweep transform record (CM3D_SWEEP_XF_STRIDE)


def quat_to_rotmat(q_wxyz):
    """Unit-quaternion (w,x,y,z) -> 3x3 float64 rotation matrix
    (pyquaternion `Quaternion(q).rotation_matrix`; the quaternion is normalised first)."""
    # |q| = sqrt(w^2 + x^2 + y^2 + z^2) with the squares added left to right, all in IEEE double on Python floats: a DEFINED
    # order (a BLAS dot product's is not), so that the native table walk (csrc/reader.cpp, cm3d_tables_*) produces the same
    # bits; this runs twenty times per frame in the entry points' Python table walk
    w, x, y, z = (float(v) for v in q_wxyz)
    nrm = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / nrm, x / nrm, y / nrm, z / nrm
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ], np.float64)


def rotmat_to_quat(R):
    """3x3 rotation -> (w,x,y,z), positive w."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def rot_z(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def scaled_intrinsic_f32(K, ratio):
    """f32(K) * f32(ratio) elementwise, K[2][2] = 1 (reference 2d_to_3d.py:585-587:
    a float32 tensor times a Python float multiplies in float32)."""
    K32 = np.asarray(K, np.float64).astype(np.float32)
    Ks = (K32 * np.float32(ratio)).astype(np.float32)
    Ks[2, 2] = np.float32(1.0)
    return Ks


def make_cam_record(stages, K_f32):
    """Camera record (float32[CAM_STRIDE], include/cm3d_hip.h): up to three rigid stages, each
    `p += t_pre; p = R p; p += t_post` with either translation optional (None), then the 3x3 K'.
    Layout: stage s at [15 s .. 15 s + 14] = t_pre(3), R(9 row-major), t_post(3); K at [45..53];
    [54] number of stages; [55] flags, bit 2s = stage s has t_pre, bit 2s+1 = stage s has t_post.
    Every entry must already be the float32 tensor the reference hands to translate/rotate/matmul."""
    if not 1 <= len(stages) <= 3:
        raise ValueError("1..3 stages")
    c = np.zeros(CAM_STRIDE, np.float32)
    flags = 0
    for s, (t_pre, Rm, t_post) in enumerate(stages):
        o = 15 * s
        if t_pre is not None:
            c[o:o + 3] = np.asarray(t_pre, np.float32).reshape(3)
            flags |= 1 << (2 * s)
        c[o + 3:o + 12] = np.asarray(Rm, np.float64).astype(np.float32).reshape(9) if np.asarray(Rm).dtype != np.float32 else np.asarray(Rm).reshape(9)
        if t_post is not None:
            c[o + 12:o + 15] = np.asarray(t_post, np.float32).reshape(3)
            flags |= 2 << (2 * s)
    c[CAM_K:CAM_K + 9] = np.asarray(K_f32, np.float32).reshape(9)
    c[54] = len(stages)
    c[55] = flags
    return c


def cam_stage(rec, s):
    """(t_pre, R, t_post) of stage s of a camera record (float64 copies; absent translations are zero)."""
    o = 15 * s
    r = np.asarray(rec, np.float64)
    return r[o:o + 3].copy(), r[o + 3:o + 12].reshape(3, 3).copy(), r[o + 12:o + 15].copy()


def cam_K(rec):
    return np.asarray(rec, np.float64)[CAM_K:CAM_K + 9].reshape(3, 3).copy()


def nusc_cam_record(ego_translation, ego_rotation_wxyz, cs_translation, cs_rotation_wxyz, K, ratio):
    """Camera record for the nuScenes chain (reference 2d_to_3d.py:569-587):
    p += f32(-t_ego); p = f32(R_ego^T) p; p += f32(-t_cs); p = f32(R_cs^T) p; K' = f32(K)*ratio."""
    return make_cam_record([((-np.asarray(ego_translation, np.float64)).astype(np.float32), quat_to_rotmat(ego_rotation_wxyz).T, None),
                            ((-np.asarray(cs_translation, np.float64)).astype(np.float32), quat_to_rotmat(cs_rotation_wxyz).T, None)],
                           scaled_intrinsic_f32(K, ratio))


def single_stage_cam_record(t_added_f32, R_f32, K_f32):
    """One rigid stage then K (Waymo chain, reference src/waymo/2d_to_3d.py:575-593)."""
    return make_cam_record([(t_added_f32, R_f32, None)], K_f32)


def sweep_xf_record(cs_translation, cs_rotation_wxyz, ego_translation, ego_rotation_wxyz):
    """Sweep transform record (reference 2d_to_3d.py:450-457):
    [0..8] f32(R_cs), [9..11] f32(t_cs), [12..20] f32(R_ego), [21..23] f32(t_ego)."""
    r = np.zeros(SWEEP_XF_STRIDE, np.float32)
    r[0:9] = quat_to_rotmat(cs_rotation_wxyz).astype(np.float32).reshape(9)
    r[9:12] = np.asarray(cs_translation, np.float64).astype(np.float32)
    r[12:21] = quat_to_rotmat(ego_rotation_wxyz).astype(np.float32).reshape(9)
    r[21:24] = np.asarray(ego_translation, np.float64).astype(np.float32)
    return r
