"""GPU parity campaign: randomly drawn batch shapes (points, sweeps, cameras, mask counts and sizes, row order, both
mask paths, distance of the ego pose from the map origin: 0 / 1.7 / 4 / 10 km) through the HIP path and the oracle, every output compared as in tests/test_gpu_parity.py.
CM3D_CAMPAIGN_SEEDS sets the number of drawn shapes (default 10; the long runs are recorded in DESIGN.md section 4),
CM3D_CAMPAIGN_FIRST moves the seed, CM3D_CAMPAIGN_MAX_SWEEPS (default 4) widens the number of sweeps per frame."""
import os

import numpy as np
import pytest

from cm3d_amd import synthetic as syn
from tests.helpers import oracle_batch
from tests.test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _draw(rng):
    w, h, ratio = [(256, 144, 0.16), (512, 288, 0.32), (1024, 576, 0.64), (1600, 900, 1.0), (333, 207, 0.2)][int(rng.integers(0, 5))]
    n_masks = int(rng.choice([1, 3, 8, 20, 31, 32, 33, 50, 70]))
    max_sweeps = int(os.environ.get("CM3D_CAMPAIGN_MAX_SWEEPS", "4"))      # 17 and more: past the fused launch's limit, the two-launch path
    return dict(n_points=int(rng.choice([64, 700, 1023, 1024, 1025, 5000, 12000, 35000])), n_sweeps=int(rng.integers(1, max_sweeps + 1)), n_masks=n_masks,
                n_cams=int(rng.integers(1, 7)), width=w, height=h, ratio=ratio, n_beams=int(rng.choice([16, 32, 64])),
                min_area=float(rng.choice([4.0, 30.0, 200.0])), max_area=float(rng.choice([400.0, 3000.0, 60000.0])),
                point_order=str(rng.choice(["ring", "firing"])), empty_mask_prob=float(rng.choice([0.0, 0.1, 0.5])),
                duplicate_prob=float(rng.choice([0.0, 0.2, 0.6])), seed=int(rng.integers(0, 1 << 30)),
                # distance of the ego pose from the map origin (None: the generator's ~1.7 km): the float32 chains and the
                # projection kernel's culling margins scale with it
                ego_magnitude=[None, 0.0, 4000.0, 10000.0][int(rng.integers(0, 4))])


def test_random_shapes_against_oracle(oracle):
    import torch
    from cm3d_amd import lifting
    n_seeds = int(os.environ.get("CM3D_CAMPAIGN_SEEDS", "10"))
    rng = np.random.default_rng(int(os.environ.get("CM3D_CAMPAIGN_FIRST", "0")) + 77)
    eng = lifting.LiftEngine(keep_colsum=False)
    tot = dict(frames=0, points=0, masks=0, hits=0, boxes=0)
    for s in range(n_seeds):
        over = _draw(rng)
        cfg = syn.config("tiny", **over)
        n_frames = int(rng.integers(1, 5))
        frames = [syn.make_frame(cfg, 1000 * s + i) for i in range(n_frames)]
        lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], int(rng.choice([1, 50, 4000])), seed=s)]
        fl = [0] * n_frames
        hb = lifting.pack_frames(frames, lanes, fl)
        eng.upload(hb)
        masks = "dense" if (s % 3 == 0 and over["width"] <= 1024) else "rle"
        if masks == "dense":
            eng.decode_masks_dense()
        eng.run(masks=masks)
        torch.cuda.synchronize()
        eng.check_status()
        got = eng.download()
        exp = oracle_batch(oracle, frames, lanes, fl, hb)
        try:
            _compare(hb, got, exp)
        except AssertionError as e:
            raise AssertionError(f"shape {s}: {over}, {n_frames} frames, masks={masks}: {e}") from e
        tot["frames"] += n_frames; tot["points"] += int(exp["pt_off"][-1]); tot["masks"] += hb.n_masks
        tot["hits"] += int(exp["hit_idx"].size); tot["boxes"] += int((exp["flags"] == 3).sum())
        if (s + 1) % 50 == 0:
            print(f"campaign: {s + 1} / {n_seeds} shapes equal so far", flush=True)     # a long run must not look hung
    print(f"campaign: {n_seeds} shapes, {tot}")
    assert tot["hits"] > 0
