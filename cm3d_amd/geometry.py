"""Host-side geometry helpers used to assemble kernel inputs.

These restate the small pieces of pyquaternion 0.9.9 / nuscenes-devkit that the
reference leans on at its boundary with the kernels (reference
src/nuscenes/2d_to_3d.py:451-457,570-577,585-587).  The kernels consume the
float32 matrices produced here, never quaternions.
"""
import numpy as np

CAM_STRIDE = 40          # floats per camera record (include/cm3d_hip.h CM3D_CAM_STRIDE)
SWEEP_XF_STRIDE = 24     # floats per sweep transform record (CM3D_SWEEP_XF_STRIDE)


def quat_to_rotmat(q_wxyz):
    """Unit-quaternion (w,x,y,z) -> 3x3 float64 rotation matrix
    (pyquaternion `Quaternion(q).rotation_matrix`; the quaternion is normalised first)."""
    q = np.asarray(q_wxyz, np.float64)
    q = q / np.linalg.norm(q)
    w, x, y, z = q
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


def nusc_cam_record(ego_translation, ego_rotation_wxyz, cs_translation, cs_rotation_wxyz, K, ratio):
    """Camera record for the nuScenes chain (reference 2d_to_3d.py:569-587):
    p += f32(-t_ego); p = f32(R_ego^T) p; p += f32(-t_cs); p = f32(R_cs^T) p; K' = f32(K)*ratio."""
    c = np.zeros(CAM_STRIDE, np.float32)
    c[0:3] = (-np.asarray(ego_translation, np.float64)).astype(np.float32)
    c[3:12] = quat_to_rotmat(ego_rotation_wxyz).T.astype(np.float32).reshape(9)
    c[12:15] = (-np.asarray(cs_translation, np.float64)).astype(np.float32)
    c[15:24] = quat_to_rotmat(cs_rotation_wxyz).T.astype(np.float32).reshape(9)
    c[24:33] = scaled_intrinsic_f32(K, ratio).reshape(9)
    c[33] = 2
    return c


def single_stage_cam_record(t_added_f32, R_f32, K_f32):
    """One rigid stage then K (Waymo chain, reference src/waymo/2d_to_3d.py:575-593)."""
    c = np.zeros(CAM_STRIDE, np.float32)
    c[0:3] = np.asarray(t_added_f32, np.float32)
    c[3:12] = np.asarray(R_f32, np.float32).reshape(9)
    c[15:24] = np.eye(3, dtype=np.float32).reshape(9)
    c[24:33] = np.asarray(K_f32, np.float32).reshape(9)
    c[33] = 1
    return c


def sweep_xf_record(cs_translation, cs_rotation_wxyz, ego_translation, ego_rotation_wxyz):
    """Sweep transform record (reference 2d_to_3d.py:450-457):
    [0..8] f32(R_cs), [9..11] f32(t_cs), [12..20] f32(R_ego), [21..23] f32(t_ego)."""
    r = np.zeros(SWEEP_XF_STRIDE, np.float32)
    r[0:9] = quat_to_rotmat(cs_rotation_wxyz).astype(np.float32).reshape(9)
    r[9:12] = np.asarray(cs_translation, np.float64).astype(np.float32)
    r[12:21] = quat_to_rotmat(ego_rotation_wxyz).astype(np.float32).reshape(9)
    r[21:24] = np.asarray(ego_translation, np.float64).astype(np.float32)
    return r
