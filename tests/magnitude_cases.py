"""Frames with rows CRAFTED onto the boundaries the projection kernel's culling relies on (tests/test_gpu_magnitude.py runs them on
the GPU; tests/test_host_logic.py checks on the CPU that they do fall on both sides)."""
import numpy as np

from cm3d_amd import geometry as geo, rle as rlemod, synthetic as syn

MAGNITUDES = [0.0, 1700.0, 4000.0, 10000.0]
RECT = (150, 90, 330, 200)
W, H = 512, 288
N_EACH = 24
N_KINDS = 10


def _rect_rle(x0, y0, x1, y1, W, H):
    ys = np.arange(y0, y1 + 1)
    return {"size": [W, H], "counts": rlemod.counts_to_string(rlemod.spans_to_counts(ys, np.full(ys.size, x0), np.full(ys.size, x1), W, H))}


def _compose(rec):
    """float64 (M, c) with p_cam = M p + c for a camera record (what wedge_setup composes on the device)."""
    M, c = np.eye(3), np.zeros(3)
    ns, fl = int(rec[54]), int(rec[55])
    for s in range(ns):
        t_pre, R, t_post = geo.cam_stage(rec, s)
        if fl & (1 << (2 * s)):
            c = c + t_pre
        M, c = R @ M, R @ c
        if fl & (2 << (2 * s)):
            c = c + t_post
    return M, c


def _craft(fr, rng, W, H, rect, n_each, mag=0.0):
    """Sensor-frame rows (n,5) float32 whose exact-arithmetic images sit on the culling boundaries of every camera:
      * u in 1 +- 0.03 and W-1 +- 0.03 (the accept limits behind the view wedge's planes u = 0, u = W), any row, depth 2.3..90 m
      * v in 1 +- 0.03 and H-1 +- 0.03
      * pixels within 0.03 of the four edges of `rect`'s eroded bounding box (the approximate projection's grown boxes)
      * depth within 2e-4 m (+ 3e-7 of the magnitude) of the minimum depth, anywhere in the image
    float32 rounding of the rows and of the kernels' own global-frame arithmetic (2.4e-4 m per ulp at 4 km) scatters them to
    both sides of each boundary: both outcomes occur, and every one must be the reference's."""
    xf = np.asarray(fr.sweep_xf[0], np.float64)
    R_cs, t_cs, R_ego, t_ego = xf[0:9].reshape(3, 3), xf[9:12], xf[12:21].reshape(3, 3), xf[21:24]
    ex0, ey0, ex1, ey1 = rect[0] + 1, rect[1] + 1, rect[2] - 1, rect[3] - 1          # after the 3x3 erosion
    rows = []
    for c in range(fr.cams.shape[0]):
        M, cv = _compose(fr.cams[c])
        K = geo.cam_K(fr.cams[c])
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        jit = lambda n: rng.uniform(-0.03, 0.03, n)
        anyu, anyv = lambda n: rng.uniform(2.0, W - 3.0, n), lambda n: rng.uniform(2.0, H - 3.0, n)
        logz = lambda n: np.exp(rng.uniform(np.log(2.31), np.log(90.0), n))
        n = n_each
        dz = 2e-4 + 3e-7 * mag        # the float32 records round the camera's position by up to an ulp of the magnitude
        targets = [
            (1.0 + jit(n), anyv(n), logz(n)), (W - 1.0 + jit(n), anyv(n), logz(n)),
            (anyu(n), 1.0 + jit(n), logz(n)), (anyu(n), H - 1.0 + jit(n), logz(n)),
            (ex0 + jit(n), rng.uniform(ey0, ey1 + 1, n), logz(n)), (ex1 + 1.0 + jit(n), rng.uniform(ey0, ey1 + 1, n), logz(n)),
            (rng.uniform(ex0, ex1 + 1, n), ey0 + jit(n), logz(n)), (rng.uniform(ex0, ex1 + 1, n), ey1 + 1.0 + jit(n), logz(n)),
            (anyu(n), anyv(n), 2.3 + rng.uniform(-dz, dz, n)),
            (rng.uniform(ex0, ex1 + 1, n), rng.uniform(ey0, ey1 + 1, n), 2.3 + rng.uniform(-dz, dz, n)),
        ]
        for u, v, z in targets:
            pc = np.stack([z * (u - cx) / fx, z * (v - cy) / fy, z], 1)
            pg = (pc - cv) @ M                                   # M^T (p_cam - c), row form
            ps = ((pg - t_ego) @ R_ego - t_cs) @ R_cs            # inverse of sensor -> ego -> global
            rows.append(ps)
    ps = np.concatenate(rows, 0).astype(np.float32)
    # a few float32 ulps of extra scatter in the sensor frame
    k = rng.integers(-4, 5, size=ps.shape).astype(np.int32)
    ps = (ps.view(np.int32) + k).view(np.float32)
    out = np.zeros((ps.shape[0], 5), np.float32)
    out[:, :3] = ps
    out[:, 3] = 7.0
    return out




def crafted_frames(mag, n_frames=3):
    """Three 512x288 frames `mag` metres from the map origin.  Per camera a mask that covers the whole image and the
    rectangle RECT; into sweep 0 go N_KINDS x N_EACH crafted rows per camera, half of them scattered one by one among the
    ordinary rows (a wave whose ONLY candidate is a boundary row), half as one block at the sweep's end (whole wave-chunks of
    boundary rows).  Returns (frames, crafted rows of each frame as (n,5) float32 in sweep 0's sensor frame)."""
    cfg = syn.config("tiny", n_points=9000, n_sweeps=2, n_masks=6, width=W, height=H, ratio=0.32, ego_magnitude=mag, point_order="firing")
    rng = np.random.default_rng(int(mag) + 5)
    frames, crafted_all = [], []
    for i in range(n_frames):
        fr = syn.make_frame(cfg, 900 + i)
        crafted = _craft(fr, rng, W, H, RECT, N_EACH, mag)
        base = fr.sweeps_raw[0]
        order = rng.permutation(crafted.shape[0])
        half = crafted.shape[0] // 2
        pos = np.sort(rng.choice(base.shape[0], half, replace=False))
        mixed = np.insert(base, pos, crafted[order[:half]], axis=0)
        fr.sweeps_raw[0] = np.ascontiguousarray(np.concatenate([mixed, crafted[order[half:]]], 0))
        for c in range(fr.cams.shape[0]):
            fr.rles.append({"size": [W, H], "counts": rlemod.counts_to_string(np.array([0, W * H], np.uint32))})      # the whole image
            fr.rles.append(_rect_rle(*RECT, W, H))
            fr.labels += ["car", "human"]; fr.scores += [0.5, 0.4]; fr.cam_nums += [c, c]
        frames.append(fr)
        crafted_all.append(crafted)
    return frames, crafted_all
