"""GPU parity across COORDINATE MAGNITUDE.  The reference's float32 chains run on global-frame coordinates, so what they
compute depends on how far the ego pose is from the map origin (nuScenes maps reach ~4 km; Waymo / KITTI clouds sit at the
origin), and the projection kernel's culling (view wedge, approximate projection against grown mask boxes, the depth it
may accept: csrc/project.hip wedge_setup) has margins that scale with that distance.  Here the whole path is held to the
oracle at 0 m, 1.7 km (the default of every other test), 4 km, 10 km and 30 km (where the approximate projection switches
itself off), on ordinary frames and on frames with rows CRAFTED to sit within a few float32 ulps of every culling
boundary: the image's left / right accept limits (floor(u) = 1, u = W - 1), the edges of a mask's bounding box, and the
minimum depth."""
import numpy as np
import pytest

from cm3d_amd import synthetic as syn
from tests.helpers import oracle_batch
from tests.magnitude_cases import MAGNITUDES, crafted_frames
from tests.test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _lift(frames, lanes, fl, oracle, **eng_kw):
    import torch
    from cm3d_amd import lifting
    hb = lifting.pack_frames(frames, lanes, fl)
    eng = lifting.LiftEngine(**eng_kw)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    return hb, eng.download(), oracle_batch(oracle, frames, lanes, fl, hb)


@pytest.mark.parametrize("mag", MAGNITUDES + [30000.0])
@pytest.mark.parametrize("shape", ["tiny", "c1", "c2"])
def test_frames_at_every_magnitude(oracle, shape, mag):
    n = {"tiny": 5, "c1": 2, "c2": 2}[shape]
    cfg = syn.config(shape, ego_magnitude=mag)
    frames = [syn.make_frame(cfg, 40 + i) for i in range(n)]
    assert abs(float(np.hypot(*frames[0].ego_xyz[:2])) - mag) < 300.0
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 4000, seed=1), syn.make_lane_table(frames[-1].ego_xyz[:2], 3000, seed=2)]
    hb, got, exp = _lift(frames, lanes, [i % 2 for i in range(n)], oracle, keep_colsum=True)
    assert exp["hit_idx"].size > (50 if shape == "tiny" else 1000)
    _compare(hb, got, exp)


@pytest.mark.parametrize("mag", MAGNITUDES + [30000.0])
def test_rows_crafted_onto_the_culling_boundaries(oracle, mag):
    """Per camera a mask that covers the whole image and a rectangle in the middle; rows crafted onto every boundary the
    culling relies on (tests/magnitude_cases.py), half of them in one block (whole wave-chunks of boundary rows), half
    scattered one by one among ordinary rows (a wave whose ONLY candidate is a boundary row).  Index lists, in-mask
    coordinates, medoids, boxes: all the oracle's.  (tests/test_host_logic.py checks on the CPU that the crafted rows fall on
    both sides of every boundary.)"""
    frames, _ = crafted_frames(mag)
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
    hb, got, exp = _lift(frames, lanes, [0] * len(frames), oracle, keep_colsum=True, keep_cloud=True)
    _compare(hb, got, exp)


def test_campaign_draws_magnitudes():
    """The randomised campaign draws the magnitude as well (tests/test_gpu_campaign.py)."""
    from tests.test_gpu_campaign import _draw
    rng = np.random.default_rng(1)
    seen = {_draw(rng)["ego_magnitude"] for _ in range(200)}
    assert {None, 0.0, 4000.0, 10000.0} <= seen
