"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
frames.  Bit-exact for indices / integer outputs and for float32 centroids; 1e-4 (north_star)
on the float64 box parameters, in practice ~1e-12."""
import os

import numpy as np
import pytest

from cm3d_amd import synthetic as syn
from tests.helpers import oracle_batch

pytestmark = pytest.mark.gpu

BOX_TOL = 1e-4    # BASELINE.json north_star tolerance on (x,y,z,l,w,h,theta)


def _run(cfg_name, n_frames, masks, oracle, keep_cloud=None, **over):
    """keep_cloud None: the dense-mask runs also materialise the cloud (and compare it), the RLE runs take the product
    default -- no cloud, in-mask coordinates re-derived from the raw rows."""
    import torch
    from cm3d_amd import lifting
    if keep_cloud is None:
        keep_cloud = masks == "dense"
    cfg = syn.config(cfg_name, **over)
    frames = [syn.make_frame(cfg, i) for i in range(n_frames)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 4000, seed=1), syn.make_lane_table(frames[-1].ego_xyz[:2], 3000, seed=2)]
    frame_lane = [i % 2 for i in range(n_frames)]
    hb = lifting.pack_frames(frames, lanes, frame_lane)
    eng = lifting.LiftEngine(keep_colsum=True, keep_cloud=keep_cloud)
    eng.upload(hb)
    if masks == "dense":
        eng.decode_masks_dense()
    eng.run(masks=masks)
    torch.cuda.synchronize()
    got = eng.download()
    assert ("points" in got) == bool(keep_cloud)
    exp = oracle_batch(oracle, frames, lanes, frame_lane, hb)
    return hb, got, exp


def _expected_hit_xyz(hb, exp):
    """Coordinates of every listed point, from the oracle's cloud: points[pt_off[frame of mask] + hit_idx]."""
    per_mask_frame = np.repeat(np.arange(hb.n_frames), np.diff(hb.mask_off))
    base = np.repeat(exp["pt_off"][per_mask_frame], np.diff(exp["hit_off"]))
    return exp["points"][base + exp["hit_idx"]]


def _compare(hb, got, exp):
    assert np.array_equal(got["pt_off"], exp["pt_off"])
    if "points" in got:
        assert np.array_equal(got["points"].view(np.uint32), exp["points"].view(np.uint32)), "sweep prep not bit-exact"
    assert np.array_equal(got["bbox"], exp["bbox"])
    assert np.array_equal(got["hit_off"], exp["hit_off"])
    assert np.array_equal(got["hit_idx"], exp["hit_idx"]), "point-to-mask index lists differ"
    assert np.array_equal(got["hit_xyz"].view(np.uint32), _expected_hit_xyz(hb, exp).view(np.uint32)), "in-mask coordinates not bit-exact"
    assert np.array_equal(got["medoid_pos"], exp["medoid_pos"])
    assert np.array_equal(got["centroid"].view(np.uint32), exp["centroid"].view(np.uint32))
    assert np.array_equal(got["lane_idx"], exp["lane_idx"])
    valid = exp["medoid_pos"] >= 0
    assert np.array_equal(got["lane_dist"][valid], exp["lane_dist"][valid])
    assert np.array_equal(got["flags"], exp["flags"])
    assert np.allclose(got["box"], exp["box"], rtol=0, atol=BOX_TOL)
    # what we actually see: ~1e-9, up to 1.2e-6 where push_centroid's 1/sin, 1/cos amplify the one-ulp difference between the
    # device's and the host's float32 cos/sin of the lane yaw (1000-shape campaign)
    assert np.abs(got["box"] - exp["box"]).max() < 1e-5


@pytest.mark.parametrize("masks", ["dense", "rle"])
def test_tiny_frames(oracle, masks):
    hb, got, exp = _run("tiny", 5, masks, oracle)
    assert exp["hit_idx"].size > 50
    _compare(hb, got, exp)


@pytest.mark.parametrize("masks", ["dense", "rle"])
def test_reference_resolution_frame(oracle, masks):
    # C1: 3 sweeps x 34.7k points, 6 x 1024x576 masks, ratio 0.64 -- the reference's own configuration
    hb, got, exp = _run("c1", 2, masks, oracle)
    assert exp["hit_idx"].size > 1000
    _compare(hb, got, exp)


def test_headline_resolution_frame(oracle):
    # C2 shape: 35k points, 6 x 1600x900 masks, ratio 1.0
    hb, got, exp = _run("c2", 2, "dense", oracle)
    _compare(hb, got, exp)


def test_colsum_bit_exact(oracle):
    hb, got, exp = _run("c1", 1, "rle", oracle)
    pts = exp["points"]
    for m in range(hb.n_masks):
        o, e = exp["hit_off"][m], exp["hit_off"][m + 1]
        if e - o == 0:
            continue
        _, cs = oracle.medoid(pts, exp["hit_idx"][o:e], want_colsum=True)
        assert np.array_equal(cs.view(np.uint32), got["colsum"][o:e].view(np.uint32)), f"mask {m} column sums differ"


def test_projection_share_of_the_chip_never_changes_a_result(oracle, raw_layout):
    """cm3d_project_workgroups_per_cu (LiftPipeline asks for 2 per CU with batches in flight) only sizes the projection launch's grid:
    the work lists hand every wave-chunk to exactly one wave whatever their number.  One batch under 0 (fill the chip), 1, 2 and 7 workgroups
    per CU: every pass the oracle's results."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("c1", n_masks=24)
    frames = [syn.make_frame(cfg, 40 + i) for i in range(3)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * 3, layout=raw_layout)
    exp = oracle_batch(oracle, frames, lanes, [0] * 3, hb)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    prev = eng.lib.cm3d_project_workgroups_per_cu(0)
    try:
        for per_cu in (0, 1, 2, 7, 0):
            eng.lib.cm3d_project_workgroups_per_cu(per_cu)
            eng.run(masks="rle")
            torch.cuda.synchronize()
            eng.check_status()
            _compare(hb, eng.download(), exp)
    finally:
        eng.lib.cm3d_project_workgroups_per_cu(prev)


def test_medoid_hint_never_changes_a_result(oracle):
    """LiftEngine feeds the medoid stage's feedback word (does this batch hold a list of more than 448 points?) back as the next
    pass's hint (cm3d_medoid2).  A batch without long lists, then -- in the same engine, so with a hint that says "none" -- a batch
    full of them, then the short one again: every pass must give the oracle's results, whatever the hint said."""
    import torch
    from cm3d_amd import lifting
    short_cfg, long_cfg = syn.config("tiny"), syn.config("c1")
    eng = lifting.LiftEngine()
    seen = []
    for cfg, seed, n in ((short_cfg, 5, 4), (long_cfg, 9, 1), (short_cfg, 6, 4)):
        frames = [syn.make_frame(cfg, seed + i) for i in range(n)]
        lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
        hb = lifting.pack_frames(frames, lanes, [0] * n)
        exp = oracle_batch(oracle, frames, lanes, [0] * n, hb)
        eng.upload(hb)
        for _ in range(3):                          # pass 1 runs on the previous batch's hint, pass 2 and 3 on this batch's
            eng.run(masks="rle")
            torch.cuda.synchronize()
            seen.append(int(eng._md_fb_np[0]))
            got = eng.download()
            _compare(hb, got, exp)
        assert (np.diff(exp["hit_off"]).max() > 448) == bool(seen[-1])          # csrc/medoid.hip MD_BATCH_LONG
    assert seen[:3] == [0, 0, 0] and seen[3:6] == [1, 1, 1] and seen[6:] == [0, 0, 0]


def test_hit_rows_of_the_byte_accounting(oracle):
    """bench.py prices the hit words of the 256-row blocks that hold an in-mask point (the only ones the projection writes and
    the compaction reads); `LiftEngine.hit_chunk_rows` reads the blocks' flags back.  Held to the oracle's index lists: compacted
    index -> raw row of the frame through the removed-row bits, rows per block of 256."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("c2")
    frames = [syn.make_frame(cfg, 70 + i) for i in range(3)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0, 0, 0])
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got_rows = eng.hit_chunk_rows()
    removed = eng.removed_rows()
    exp = oracle_batch(oracle, frames, lanes, [0, 0, 0], hb)
    p_off = eng.b.pt_off.cpu().numpy()
    want = 0
    for f in range(3):
        kept = np.flatnonzero(~removed[p_off[f]:p_off[f + 1]])           # raw row of every row of the reference's cloud
        n = int(p_off[f + 1] - p_off[f])
        blocks = set()
        for m in range(hb.mask_off[f], hb.mask_off[f + 1]):
            blocks.update((kept[exp["hit_idx"][exp["hit_off"][m]:exp["hit_off"][m + 1]]] // 256).tolist())
        want += sum(min(256, n - 256 * c) for c in blocks)
    assert 0 < want < int(p_off[-1]) and got_rows == want


def test_waymo_shaped_frame(oracle):
    # C4 shape at reduced point count: 5 cameras, 1920x1280 masks, 64 beams
    hb, got, exp = _run("c4", 1, "rle", oracle, n_points=60000)
    assert exp["hit_idx"].size > 500
    _compare(hb, got, exp)


def test_many_masks_three_planes(oracle):
    # C5 density: 80 masks per frame -> 3 hit-word planes; 2 sweeps
    hb, got, exp = _run("c5", 2, "rle", oracle, n_sweeps=2, n_points=20000, width=1024, height=576, ratio=0.64)
    assert hb.n_masks == 160
    _compare(hb, got, exp)


def test_a_thousand_masks_in_a_frame(oracle):
    """The per-frame mask limit (CM3D_MAX_MASKS_PER_FRAME = 1024, 32 hit-word planes): 1000 small masks in one frame."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny", n_masks=1000, n_points=6000, duplicate_prob=0.5)
    frames = [syn.make_frame(cfg, 3), syn.make_frame(syn.config("tiny"), 4)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 1000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0, 0])
    assert hb.mask_off[1] == 1000
    eng = lifting.LiftEngine(keep_colsum=True)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, [0, 0], hb)
    assert exp["hit_idx"].size > 5000
    _compare(hb, got, exp)


def test_many_small_frames(oracle):
    """1200 small frames in one batch (grid dimensions, per-frame workgroups, offsets across many frames)."""
    hb, got, exp = _run("tiny", 1200, "rle", oracle)
    assert exp["hit_idx"].size > 50000
    _compare(hb, got, exp)


def test_ragged_batch(oracle):
    """Frames with different point counts and mask counts in one batch, incl. a frame whose masks are all empty."""
    import torch
    from cm3d_amd import lifting, rle
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(4)]
    frames[1].sweeps_raw = [frames[1].sweeps_raw[0][:700]]
    frames[1].sweep_xf = frames[1].sweep_xf[:1]
    frames[2].rles = frames[2].rles[:3]; frames[2].labels = frames[2].labels[:3]
    frames[2].scores = frames[2].scores[:3]; frames[2].cam_nums = frames[2].cam_nums[:3]
    empty = rle.counts_to_string(np.array([cfg.width * cfg.height], np.uint32))
    frames[3].rles = [{"size": [cfg.width, cfg.height], "counts": empty} for _ in frames[3].rles]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 1000, seed=1)]
    fl = [0, 0, 0, 0]
    hb = lifting.pack_frames(frames, lanes, fl)
    eng = lifting.LiftEngine(keep_colsum=True)
    eng.upload(hb)
    for masks in ("rle", "dense"):
        if masks == "dense":
            eng.decode_masks_dense()
        eng.run(masks=masks)
        torch.cuda.synchronize()
        got = eng.download()
        exp = oracle_batch(oracle, frames, lanes, fl, hb)
        _compare(hb, got, exp)
    m0 = hb.mask_off[3]
    assert (got["flags"][m0:] == 0).all() and (got["medoid_pos"][m0:] == -1).all()


def test_heavy_ego_box_drop(oracle):
    """A third of the rows of every sweep fall inside the ego box (:442-445), scattered over all workgroups of the
    frame: the index lists must still be the indices of the reference's compacted cloud."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("c1")
    frames = [syn.make_frame(cfg, i) for i in range(2)]
    rng = np.random.default_rng(5)
    for fr in frames:
        for sw in fr.sweeps_raw:
            sel = rng.random(sw.shape[0]) < 0.33
            sw[sel, 0] = rng.uniform(-1.5, 1.5, int(sel.sum())).astype(np.float32)
            sw[sel, 1] = rng.uniform(-1.5, 1.5, int(sel.sum())).astype(np.float32)
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 2000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0, 0])
    eng = lifting.LiftEngine(keep_colsum=True)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, [0, 0], hb)
    assert exp["pt_off"][-1] < 0.75 * hb.n_raw_rows and exp["hit_idx"].size > 500
    _compare(hb, got, exp)


def test_index_capacity_overflow_is_reported(oracle):
    """Too small an index buffer: nothing is written out of bounds, the status word says so and the host raises."""
    import torch
    from cm3d_amd import _lib, lifting
    cfg = syn.config("c1")
    frames = [syn.make_frame(cfg, i) for i in range(2)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 2000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0, 0])
    eng = lifting.LiftEngine()
    eng.hits_per_point = 0.0                 # -> the minimum capacity of 1024 indices, far below this batch's hits
    eng.upload(hb)
    guard = torch.full((4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")     # allocated right after the engine's buffers
    eng.run(masks="rle")
    torch.cuda.synchronize()
    with pytest.raises(_lib.Cm3dError, match="capacity"):
        eng.download()
    assert int((guard != 0x5A5A5A5A).sum()) == 0
    # the same engine recovers with a sufficient capacity
    eng.hits_per_point = 2.0
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, [0, 0], hb)
    _compare(hb, got, exp)


def test_frame_without_points(oracle):
    """A frame whose only sweep is empty (and one whose rows all fall into the ego box): no points, no boxes, and the
    neighbouring frames are unaffected."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(4)]
    frames[1].sweeps_raw = [frames[1].sweeps_raw[0][:0]]
    frames[1].sweep_xf = frames[1].sweep_xf[:1]
    for sw in frames[2].sweeps_raw:
        sw[:, 0] = 0.5; sw[:, 1] = -0.5          # every row inside the ego box
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 1000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * 4)
    eng = lifting.LiftEngine(keep_colsum=True)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, [0] * 4, hb)
    _compare(hb, got, exp)
    for f in (1, 2):
        m0, m1 = hb.mask_off[f], hb.mask_off[f + 1]
        assert got["pt_off"][f + 1] == got["pt_off"][f] and (got["flags"][m0:m1] == 0).all()
    assert (got["flags"][:hb.mask_off[1]] != 0).any() and (got["flags"][hb.mask_off[3]:] != 0).any()


_SPLIT_SCRIPT = """
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from cm3d_amd import lifting, synthetic as syn
cfg = syn.config("tiny", n_points=9000, n_sweeps=3, n_masks=40, width=512, height=288, ratio=0.32)
frames = [syn.make_frame(cfg, 70 + i) for i in range(5)]
lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 2000, seed=1)]
eng = lifting.LiftEngine()
eng.upload(lifting.pack_frames(frames, lanes, [0] * 5))
eng.run(masks="rle")
torch.cuda.synchronize()
eng.check_status()
got = eng.download()
h = hashlib.sha256()
for k in sorted(got):
    h.update(k.encode()); h.update(np.ascontiguousarray(got[k]).tobytes())
print("DIGEST", h.hexdigest(), int(got["hit_off"][-1]))
"""


def test_work_split_of_the_projection_changes_no_result():
    """How the projection launch deals its wave-chunks out (tickets per frame, workgroups, who steals what) must not show in any
    output: the same batch under very different splits -- one list per frame, odd list counts, far more lists than resident
    waves, a launch of eight workgroups -- gives byte-identical downloads."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for env in ({}, {"CM3D_PH_TICKETS": "1"}, {"CM3D_PH_TICKETS": "7"}, {"CM3D_PH_TICKETS": "18"}, {"CM3D_PH_BLOCKS": "8"}):
        r = subprocess.run([sys.executable, "-c", _SPLIT_SCRIPT.format(root=root)], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0].split()
        assert int(line[2]) > 1000                      # the batch has in-mask points at all
        digests.append(line[1])
    assert len(set(digests)) == 1, digests


def test_second_pass_is_identical(oracle):
    """Running the resident batch twice gives bit-identical outputs (no state leaks between passes,
    no order-dependent atomics on anything that is an output)."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(6)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 2000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * 6)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    outs = []
    for _ in range(3):
        eng.run(masks="rle")
        torch.cuda.synchronize()
        outs.append(eng.download())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True) and np.array_equal(outs[0][k], outs[2][k], equal_nan=True), k


def test_wider_sample_against_oracle(oracle):
    """A dozen headline-shaped frames (C2, firing-order sweeps) against the oracle: ~400 k points x 6 cameras through
    the packed-float32 projection and its division shortcut, every index list compared."""
    hb, got, exp = _run("c2", 12, "rle", oracle)
    assert exp["hit_idx"].size > 10000
    _compare(hb, got, exp)


def _check_batch_properties(hb, a):
    """Size-independent properties of one downloaded pass (shapes too large for the oracle)."""
    F = hb.n_frames
    off, idx = a["hit_off"], a["hit_idx"]
    assert off[0] == 0 and off[-1] == idx.size and np.all(np.diff(off) >= 0)
    # ascending inside every list, inside the frame's point range
    starts = np.zeros(idx.size, bool); starts[off[:-1][np.diff(off) > 0]] = True
    assert np.all((np.diff(idx) > 0) | starts[1:])
    n_pts = np.diff(a["pt_off"])
    per_mask_frame = np.repeat(np.arange(F), np.diff(hb.mask_off))
    cnt = np.diff(off)
    assert np.all(idx >= 0) and np.all(idx < np.repeat(n_pts[per_mask_frame], cnt))
    # the reference drops the ego-box rows: what is left is what the frame's sweeps hold minus those rows
    raw_per_frame = np.diff(hb.sweep_row_off[hb.frame_sweep_off])
    assert np.all(n_pts <= raw_per_frame) and (not hb.ego_box or n_pts.sum() < raw_per_frame.sum())
    # medoid position inside its list, exactly the masks with points have a centroid
    mp = a["medoid_pos"]
    assert np.array_equal(mp >= 0, cnt > 0) and np.all(mp[cnt > 0] < cnt[cnt > 0])
    # the centroid is the medoid point of the mask's in-mask points, and those are finite global-frame coordinates
    sel = np.nonzero(cnt > 0)[0]
    pts = a["hit_xyz"][off[sel] + mp[sel]]
    assert np.array_equal(pts[:, :3].view(np.uint32), a["centroid"][sel].view(np.uint32))
    assert np.all(np.isfinite(a["hit_xyz"][:, :3]))
    # a box exists (bit 0) exactly for the masks with points; NMS (bit 1) keeps a subset of them
    assert np.array_equal((a["flags"] & 1) != 0, cnt > 0) and np.all(((a["flags"] & 2) == 0) | (cnt > 0))
    return cnt


def test_full_size_batch_properties():
    """BASELINE's full C2 batch (256 frames x 35 k points x 20 masks of 1600x900), too large for the oracle: the two
    independent mask paths (run lengths -> packed, and run lengths -> dense bytes -> packed) must give identical
    results, index lists must be ascending and in range, counts must add up, and a second pass must reproduce
    the first bit for bit."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("c2")
    F = 256
    frames = [syn.make_frame(cfg, 5000 + i) for i in range(F)]
    lanes = [syn.make_lane_table([600.0, 1600.0], 50000, seed=3, extent=260.0)]
    hb = lifting.pack_frames(frames, lanes, [0] * F)
    del frames
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    a = eng.download()
    eng.run(masks="rle")
    torch.cuda.synchronize()
    a2 = eng.download()
    eng.decode_masks_dense()
    eng.run(masks="dense")
    torch.cuda.synchronize()
    b = eng.download()
    for k in a:
        assert np.array_equal(a[k], a2[k], equal_nan=True), f"second pass differs in {k}"
        assert np.array_equal(a[k], b[k], equal_nan=True), f"mask paths differ in {k}"
    _check_batch_properties(hb, a)
    assert a["hit_idx"].size > 200000 and int(((a["flags"] & 3) == 3).sum()) > 2000
    # with the cloud materialised (keep_cloud) every result is the same, and the cloud's rows are the listed coordinates
    eng2 = lifting.LiftEngine(keep_cloud=True)
    eng2.upload(hb)
    eng2.run(masks="rle")
    torch.cuda.synchronize()
    c = eng2.download()
    for k in a:
        assert np.array_equal(a[k], c[k], equal_nan=True), f"keep_cloud changes {k}"
    per_mask_frame = np.repeat(np.arange(F), np.diff(hb.mask_off))
    base = np.repeat(c["pt_off"][per_mask_frame], np.diff(c["hit_off"]))
    assert np.array_equal(c["points"][base + c["hit_idx"]].view(np.uint32), c["hit_xyz"].view(np.uint32))


def _full_size(cfg_name, n_batch, oracle):
    """One BASELINE shape at full size: frame 0 against the oracle (everything _compare checks), a batch of n_batch frames
    through the size-independent properties, and a second pass that must reproduce the first bit for bit."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config(cfg_name)
    frames = [syn.make_frame(cfg, 7000 + i) for i in range(n_batch)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 50000, seed=5, extent=400.0)]
    hb = lifting.pack_frames(frames, lanes, [0] * n_batch)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    a = eng.download()
    eng.run(masks="rle")
    torch.cuda.synchronize()
    a2 = eng.download()
    for k in a:
        assert np.array_equal(a[k], a2[k], equal_nan=True), f"second pass differs in {k}"
    cnt = _check_batch_properties(hb, a)
    # frame 0 alone, against the oracle
    hb1 = lifting.pack_frames(frames[:1], lanes, [0])
    eng1 = lifting.LiftEngine(keep_colsum=True, keep_cloud=True)
    eng1.upload(hb1)
    eng1.run(masks="rle")
    torch.cuda.synchronize()
    got = eng1.download()
    exp = oracle_batch(oracle, frames[:1], lanes, [0], hb1)
    _compare(hb1, got, exp)
    # ... and frame 0 inside the batch gives what it gives alone
    m1 = hb1.n_masks
    assert np.array_equal(a["hit_off"][:m1 + 1], got["hit_off"]) and np.array_equal(a["hit_idx"][:got["hit_idx"].size], got["hit_idx"])
    assert np.array_equal(a["medoid_pos"][:m1], got["medoid_pos"]) and np.array_equal(a["flags"][:m1], got["flags"])
    return hb, a, cnt


def test_c4_waymo_shape_at_full_size(oracle):
    """BASELINE C4: 180 k points, 5 cameras, 20 masks of 1920x1280 per frame."""
    hb, a, cnt = _full_size("c4", 12, oracle)
    assert hb.n_real_rows == 12 * 180000 and (hb.width, hb.height, hb.n_cams) == (1920, 1280, 5)
    assert a["hit_idx"].size > 20000


def test_c5_ten_sweeps_eighty_masks_at_full_size(oracle):
    """BASELINE C5: 10 sweeps x 35 k points and 80 masks of 1600x900 per frame (3 hit-word planes, lists of thousands of
    points: the two-pass medoid on the batch, the exact one -- keep_colsum -- on the frame compared with the oracle)."""
    hb, a, cnt = _full_size("c5", 6, oracle)
    assert hb.n_real_rows == 6 * 350000 and int(np.diff(hb.mask_off).max()) == 80 and int(np.diff(hb.frame_sweep_off).max()) == 10
    assert cnt.max() > 512 and a["hit_idx"].size > 100000


def test_waymo_pipeline(oracle):
    """a17: single-stage cameras, no ego-box filter, medoid -> global for the lane lookup, boxes back in the
    vehicle frame with heading, NMS per Waymo type.  The float32 pose inverse limits agreement with the
    float64 truth to ~1e-3 m in the reference itself; against the oracle (same inputs) we still see ~1e-9."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny", n_cams=5)
    frames = [syn.make_waymo_frame(cfg, i) for i in range(4)]
    centers = [np.asarray(f.pose).reshape(4, 4)[:2, 3] for f in frames]
    lanes = [syn.make_lane_table(centers[0], 30000, seed=4, extent=400.0)]
    fl = [0] * len(frames)
    classes = lifting.ClassTable.waymo()
    hb = lifting.pack_frames(frames, lanes, fl, classes)
    assert hb.pose_rt is not None and not hb.ego_box
    eng = lifting.LiftEngine(classes=classes, keep_colsum=True)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, fl, hb)
    assert exp["hit_idx"].size > 100 and (exp["flags"] == 3).sum() > 5
    _compare(hb, got, exp)
    # the writer produces a parseable length-delimited stream with one Object per kept box
    from cm3d_amd import waymo as wm
    objs = wm.objects_from_results(hb, got, classes, [(f.context_name, f.timestamp_micros) for f in frames])
    blob = wm.encode_objects(objs)
    assert len(objs) == int((got["flags"] == 3).sum()) and blob[:1] == b"\x0a" and len(blob) > 50 * len(objs)


def test_kitti_pipeline(oracle):
    """a18: velo -> ref cloud, 3-stage camera chain (ref -> velo -> ref -> rect), single camera, no ego-box
    filter; index lists and medoids bit-exact; label lines follow save_pred."""
    import torch
    from cm3d_amd import kitti as kt, lifting
    cfg = syn.config("tiny", width=320, height=96, ratio=0.2, n_masks=10)
    frames = [syn.make_kitti_frame(cfg, i)[0] for i in range(3)]
    lanes = [np.zeros((1, 3))]
    hb = lifting.pack_frames(frames, lanes, [0, 0, 0])
    assert not hb.ego_box and hb.cams[0, 0, 54] == 3
    eng = lifting.LiftEngine(keep_colsum=True)
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = eng.download()
    exp = oracle_batch(oracle, frames, lanes, [0, 0, 0], hb)
    assert exp["hit_idx"].size > 50
    for k in ("pt_off", "hit_off", "hit_idx", "medoid_pos", "bbox"):
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(got["hit_xyz"].view(np.uint32), _expected_hit_xyz(hb, exp).view(np.uint32))
    assert np.array_equal(got["centroid"].view(np.uint32), exp["centroid"].view(np.uint32))
    exp["hit_xyz"] = _expected_hit_xyz(hb, exp)
    pri = lifting.SHAPE_PRIORS_CHATGPT
    pred, pseudo = kt.labels_of_frame(hb, got, 0, lifting.ClassTable.nuscenes(), pri)
    pred_e, _ = kt.labels_of_frame(hb, exp, 0, lifting.ClassTable.nuscenes(), pri)
    assert pred == pred_e and len(pred) == int((np.diff(exp["hit_off"][hb.mask_off[0]:hb.mask_off[1] + 1]) > 3).sum())
    f = pred[0].split()
    assert len(f) == 16 and f[1:8] == ["-1", "-1", "-10", "0", "0", "0", "0"] and len(pseudo[0].split()) == 15


def test_graph_replay_equals_eager(oracle):
    """One pass captured into a HIP graph (LiftEngine.capture_graph) gives the eager results, replay after replay."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(4)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 2000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * 4)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    eager = eng.download()
    g = eng.capture_graph(masks="rle")
    for _ in range(3):
        eng.b.hit_idx.fill_(-7); eng.b.box.fill_(0)          # whatever the replay does not rewrite would show
        g.replay()
        torch.cuda.synchronize()
        got = eng.download()
        for k in eager:
            assert np.array_equal(eager[k], got[k], equal_nan=True), k
    exp = oracle_batch(oracle, frames, lanes, [0] * 4, hb)
    _compare(hb, got, exp)


def test_pipeline_slots_and_unfused_path(oracle):
    """LiftPipeline (several batches in flight on their own streams, slots reused) gives every batch the oracle's
    results; the separate sweep + projection launches (CM3D_FUSED_SWEEPS=0, or more than 16 sweeps in a frame) give
    exactly what the fused launch gives."""
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    pipe = lifting.LiftPipeline("cuda:0", depth=2)
    batches, slots = [], []
    for k in range(5):
        frames = [syn.make_frame(cfg, 100 * k + i) for i in range(2 + k % 3)]
        lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=k)]
        fl = [0] * len(frames)
        hb = lifting.pack_frames(frames, lanes, fl)
        if len(slots) == pipe.depth:
            s0, (f0, l0, fl0, hb0) = slots.pop(0), batches.pop(0)
            got_hb, got = pipe.collect(s0)
            assert got_hb is hb0
            _compare(hb0, got, oracle_batch(oracle, f0, l0, fl0, hb0))
        slots.append(pipe.submit(hb, "rle"))
        batches.append((frames, lanes, fl, hb))
    while slots:
        s0, (f0, l0, fl0, hb0) = slots.pop(0), batches.pop(0)
        _, got = pipe.collect(s0)
        _compare(hb0, got, oracle_batch(oracle, f0, l0, fl0, hb0))
    # fused against separate launches, and the many-sweeps fallback against the oracle
    cfg3 = syn.config("tiny", n_sweeps=3, n_points=2500)
    frames = [syn.make_frame(cfg3, i) for i in range(3)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * 3)
    outs = []
    for fused in (True, False):
        eng = lifting.LiftEngine()
        eng.fused_sweeps = fused
        eng.upload(hb)
        assert eng.can_fuse_sweeps() == fused
        eng.run(masks="rle")
        torch.cuda.synchronize()
        outs.append(eng.download())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), k
    _compare(hb, outs[0], oracle_batch(oracle, frames, lanes, [0] * 3, hb))
    cfg17 = syn.config("tiny", n_sweeps=17, n_points=400)
    frames = [syn.make_frame(cfg17, i) for i in range(2)]
    hb = lifting.pack_frames(frames, lanes, [0] * 2)
    eng = lifting.LiftEngine()
    eng.fused_sweeps = True              # (whatever CM3D_FUSED_SWEEPS says) 17 sweeps in a frame: the engine must fall back
    eng.upload(hb)
    assert not eng.can_fuse_sweeps()
    eng.run(masks="rle")
    torch.cuda.synchronize()
    _compare(hb, eng.download(), oracle_batch(oracle, frames, lanes, [0] * 2, hb))
    cfg16 = syn.config("tiny", n_sweeps=16, n_points=400)
    frames = [syn.make_frame(cfg16, i) for i in range(2)]
    hb = lifting.pack_frames(frames, lanes, [0] * 2)
    eng.upload(hb)
    assert eng.can_fuse_sweeps()
    eng.run(masks="rle")
    torch.cuda.synchronize()
    _compare(hb, eng.download(), oracle_batch(oracle, frames, lanes, [0] * 2, hb))


def test_long_lists_two_pass_medoid(oracle):
    """Lists of more than 512 in-mask points take the two-pass medoid (approximate column sums, then exact sums of the few
    columns that can still be the minimum): positions and centroids must stay bit-identical to the oracle's."""
    hb, got, exp = _run("tiny", 2, "rle", oracle, n_points=30000, n_sweeps=5, n_masks=14, width=512, height=288, ratio=0.32,
                        min_area=6000.0, max_area=40000.0, empty_mask_prob=0.0, duplicate_prob=0.3, point_order="firing")
    sizes = np.diff(exp["hit_off"])
    assert (sizes > 512).sum() >= 4 and sizes.max() > 1500, sizes
    assert "colsum" in got            # _run keeps the column sums -> exact path; now the default engine
    _compare(hb, got, exp)
    import torch
    from cm3d_amd import lifting
    cfg = syn.config("tiny", n_points=30000, n_sweeps=5, n_masks=14, width=512, height=288, ratio=0.32, min_area=6000.0,
                     max_area=40000.0, empty_mask_prob=0.0, duplicate_prob=0.3, point_order="firing")
    frames = [syn.make_frame(cfg, i) for i in range(2)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 4000, seed=1), syn.make_lane_table(frames[-1].ego_xyz[:2], 3000, seed=2)]
    hb2 = lifting.pack_frames(frames, lanes, [0, 1])
    eng = lifting.LiftEngine()          # no colsum -> two-pass route for the long lists
    eng.upload(hb2)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got2 = eng.download()
    assert np.array_equal(got2["medoid_pos"], exp["medoid_pos"])
    assert np.array_equal(got2["centroid"].view(np.uint32), exp["centroid"].view(np.uint32))
    _compare(hb2, got2, exp)
