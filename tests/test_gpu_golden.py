"""GPU: each kernel, called through the C-ABI via the reference-named mirrors in cm3d_amd.ops,
against the golden vectors frozen from the reference's own helpers (tests/golden/)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_g2_index_lists_from_reference_body():
    from cm3d_amd import ops
    g = np.load(os.path.join(G, "g2_index_lists.npz"))
    W, H = (int(v) for v in g["wh"])
    for k in (0, 1):
        off, ioff = g[f"rle_off{k}"], g[f"idx_off{k}"]
        n = len(g[f"cam_nums{k}"])
        counts = [g[f"rle_counts{k}"][off[m]:off[m + 1]] for m in range(n)]
        for path in ("dense", "rle"):
            if path == "dense":
                dense = ops.decode([{"size": [W, H], "counts": c} for c in counts], as_counts=True)
                packed, bbox = ops.erode(dense)
            else:
                packed, bbox = ops.erode_rle(counts, W, H)
            lists = ops.points_in_masks(g[f"pts{k}"], g[f"cams{k}"], packed, bbox, g[f"cam_nums{k}"], W, H)
            for m in range(n):
                assert np.array_equal(lists[m], g[f"idx{k}"][ioff[m]:ioff[m + 1]]), f"frame {k} mask {m} ({path})"


@pytest.mark.parametrize("fixture", ["g2b_c1_frame.npz", "g2c_c2_frame.npz"])
def test_g2b_reference_resolution_frame_index_lists(fixture):
    """G2 at the reference's own configuration (104 k points, 1024x576 masks at ratio 0.64, 24 masks) and at the headline
    one (35 k points, 20 masks of 1600x900): the whole-frame kernel path -- fused sweep preparation, projection, compaction --
    against the index lists the reference's loop body produced."""
    import torch
    from cm3d_amd import lifting, synthetic as syn
    from tests.test_oracle_golden import _g2b_frame
    cfg, f, P, g = _g2b_frame(fixture)
    lanes = [syn.make_lane_table(f.ego_xyz[:2], 2000, seed=1)]
    hb = lifting.pack_frames([f], lanes, [0])
    for masks in ("rle", "dense"):
        eng = lifting.LiftEngine(keep_cloud=masks == "dense")
        eng.upload(hb)
        if masks == "dense":
            eng.decode_masks_dense()
        eng.run(masks=masks)
        torch.cuda.synchronize()
        got = eng.download()
        assert np.array_equal(got["hit_off"], g["idx_off"]) and np.array_equal(got["hit_idx"], g["idx"]), masks
        assert np.array_equal(got["hit_xyz"].view(np.uint32), P[g["idx"]].view(np.uint32))


def test_g3b_medoid_on_real_in_mask_lists():
    """The reference's get_medoid on 351 real in-mask lists (global-frame magnitudes, a third with duplicated rows, up to
    2284 points): the kernel -- all lists in ONE call, laid out like the compaction's hit_xyz, two-pass route for the long
    ones -- returns the reference's index on every list; the row-gather form agrees on a sample."""
    import torch
    from cm3d_amd import _lib, ops
    g = np.load(os.path.join(G, "g3b_medoid_lists.npz"))
    off = g["off"].astype(np.int32)
    n, tot = len(off) - 1, int(off[-1])
    L = _lib.lib()
    dev = torch.device("cuda:0")
    xyz = torch.zeros(tot, 4, dtype=torch.float32, device=dev)
    xyz[:, :3] = torch.from_numpy(g["pts"]).to(dev)
    hit_off = torch.from_numpy(off).to(dev)
    tiles = (np.diff(off) + _lib.MEDOID_TILE - 1) // _lib.MEDOID_TILE
    tile_off = torch.from_numpy(np.concatenate([[0], np.cumsum(tiles)]).astype(np.int32)).to(dev)
    med, cen = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.cm3d_medoid_workspace_bytes(n, tot)), dtype=torch.uint8, device=dev)
    _lib.check(L.cm3d_medoid(xyz.data_ptr(), 0, 0, n, hit_off.data_ptr(), tile_off.data_ptr(), 0, tot, 0, med.data_ptr(), cen.data_ptr(), 0,
                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "cm3d_medoid")
    got = med.cpu().numpy()
    assert np.array_equal(got, g["ref_index"]), np.flatnonzero(got != g["ref_index"])
    assert np.array_equal(cen.cpu().numpy().view(np.uint32), g["pts"][off[:-1] + g["ref_index"]].view(np.uint32))
    for k in list(range(0, n, 17)) + [int(np.argmax(np.diff(off)))]:
        p = g["pts"][off[k]:off[k + 1]]
        assert ops.get_medoid(p.T, via_rows=True) == int(g["ref_index"][k]) == ops.get_medoid(p.T), k


def test_g2d_index_lists_at_0_4_and_10_km():
    """G2 with the ego pose 0 m / 4 km / 10 km from the map origin: the whole-frame kernel path (culling by view wedge and
    approximate projection included -- their margins scale with the magnitude) against the reference's loop body."""
    import torch
    from cm3d_amd import lifting, synthetic as syn
    from tests.test_oracle_golden import _g2d_frames
    for mag, cfg, f, P, idx, off in _g2d_frames():
        lanes = [syn.make_lane_table(f.ego_xyz[:2], 2000, seed=1)]
        hb = lifting.pack_frames([f], lanes, [0])
        for masks in ("rle", "dense"):
            eng = lifting.LiftEngine(keep_cloud=masks == "dense")
            eng.upload(hb)
            if masks == "dense":
                eng.decode_masks_dense()
            eng.run(masks=masks)
            torch.cuda.synchronize()
            got = eng.download()
            assert np.array_equal(got["hit_off"], off) and np.array_equal(got["hit_idx"], idx), (mag, masks)
            assert np.array_equal(got["hit_xyz"].view(np.uint32), P[idx].view(np.uint32)), (mag, masks)


def test_g3c_medoid_at_0_4_and_10_km():
    """The reference's get_medoid on real in-mask lists at 0 m / 4 km / 10 km, all lists in one call (two-pass route, matrix
    pipe first pass for the long ones): the reference's index on every list."""
    import torch
    from cm3d_amd import _lib
    g = np.load(os.path.join(G, "g3c_medoid_magnitude.npz"))
    off = g["off"].astype(np.int32)
    n, tot = len(off) - 1, int(off[-1])
    L = _lib.lib()
    dev = torch.device("cuda:0")
    xyz = torch.zeros(tot, 4, dtype=torch.float32, device=dev)
    xyz[:, :3] = torch.from_numpy(g["pts"]).to(dev)
    hit_off = torch.from_numpy(off).to(dev)
    tiles = (np.diff(off) + _lib.MEDOID_TILE - 1) // _lib.MEDOID_TILE
    tile_off = torch.from_numpy(np.concatenate([[0], np.cumsum(tiles)]).astype(np.int32)).to(dev)
    med, cen = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.cm3d_medoid_workspace_bytes(n, tot)), dtype=torch.uint8, device=dev)
    _lib.check(L.cm3d_medoid(xyz.data_ptr(), 0, 0, n, hit_off.data_ptr(), tile_off.data_ptr(), 0, tot, 0, med.data_ptr(), cen.data_ptr(), 0,
                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "cm3d_medoid")
    got = med.cpu().numpy()
    bad = np.flatnonzero(got != g["ref_index"])
    assert bad.size == 0, [(int(k), float(g["ego_magnitude"][k]), int(off[k + 1] - off[k])) for k in bad]
    assert np.array_equal(cen.cpu().numpy().view(np.uint32), g["pts"][off[:-1] + g["ref_index"]].view(np.uint32))


def test_g3_get_medoid():
    from cm3d_amd import ops
    cases = json.load(open(os.path.join(G, "g3_medoid.json")))
    g = np.load(os.path.join(G, "g3_medoid.npz"))
    for c in cases:
        assert ops.get_medoid(g[c["name"]].T) == c["ref_index"], c["name"]


def test_g3_column_sums_equal_oracle(oracle):
    from cm3d_amd import ops
    g = np.load(os.path.join(G, "g3_medoid.npz"))
    for name in ("M25_global", "M26_global", "M300_local", "M2000_global"):
        p = g[name]
        _, cs = ops.get_medoid(p.T, want_colsum=True)
        P4 = np.concatenate([p, np.zeros((p.shape[0], 1), np.float32)], 1)
        _, exp = oracle.medoid(P4, np.arange(p.shape[0]), want_colsum=True)
        assert np.array_equal(cs.view(np.uint32), exp.view(np.uint32)), name


def test_g4_push_centroid():
    from cm3d_amd import ops
    from cm3d_amd.lifting import ClassTable
    ct = ClassTable.nuscenes()
    cases = json.load(open(os.path.join(G, "g4_push_centroid.json")))
    names = ["car", "truck", "bus", "trailer", "construction_vehicle", "pedestrian", "motorcycle", "bicycle", "traffic_cone", "barrier"]
    worst = 0.0
    for c in cases[::3]:
        name = names[c["class"]]
        t, q = ops.push_centroid(np.float32(c["centroid"]), name, np.float32(c["yaw"]), {"translation": c["ego"]}, ct)
        if name in ("car", "truck", "bus", "trailer", "construction_vehicle", "barrier"):
            worst = max(worst, np.abs(t - c["pushed"]).max(), np.abs(q - c["quat_wxyz"]).max())
        else:       # not a pushed class: raw medoid, identity rotation (2d_to_3d.py:802-806)
            assert np.array_equal(t, np.float64(np.float32(c["centroid"]))) and q.tolist() == [1, 0, 0, 0]
    assert worst < 1e-4        # north_star tolerance; we see ~5e-7 (float32 cos/sin differences)


def test_g5_circle_nms():
    from cm3d_amd import ops
    from cm3d_amd.lifting import THRESHS_BY_LABEL
    g = json.load(open(os.path.join(G, "g5_circle_nms.json")))
    for c in g["reference_cases"] + [g["tie_case_pinned"]]:
        dets = np.concatenate([np.array(c["xy"]), np.array(c["scores"])[:, None]], 1)
        assert ops.circle_nms(dets, c["labels"], THRESHS_BY_LABEL) == c["keep"]


def test_g6_lane_yaws_distances_and_coords():
    from cm3d_amd import ops
    g = np.load(os.path.join(G, "g6_lane_nn.npz"))
    yaws, dists, coords = ops.lane_yaws_distances_and_coords(g["centroids"], g["lane"])
    assert np.array_equal(yaws, g["yaws"]) and np.array_equal(dists, g["dists"]) and np.array_equal(coords, g["coords"])


def test_erode_and_decode_kernels_equal_oracle(oracle):
    from cm3d_amd import ops, rle
    rng = np.random.default_rng(5)
    for (W, H) in [(64, 40), (100, 33), (1024, 576), (1600, 900)]:
        masks = []
        for _ in range(3):
            m = np.zeros((H, W), np.uint8)
            for _ in range(6):
                x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
                m[max(0, y0 - rng.integers(1, H // 3)):y0 + rng.integers(1, H // 3), max(0, x0 - rng.integers(1, W // 3)):x0 + rng.integers(1, W // 3)] = 1
            masks.append(m)
        masks.append(np.ones((H, W), np.uint8))
        masks.append(np.zeros((H, W), np.uint8))
        edge = np.zeros((H, W), np.uint8); edge[:, :3] = 1; edge[:2, :] = 1; edge[:, -2:] = 1
        masks.append(edge)
        # long run lists (streamed 2048 runs at a time): dense noise that still leaves eroded pixels, stripes
        masks.append((rng.random((H, W)) < 0.97).astype(np.uint8))
        stripes = np.zeros((H, W), np.uint8); stripes[:, (np.arange(W) % 7) < 4] = 1; stripes[H // 2:, :] ^= 1
        masks.append(stripes)
        counts = [rle.dense_to_counts(m) for m in masks]
        assert W * H < 30000 or max(len(c) for c in counts) > 2048
        dense = ops.decode([{"size": [W, H], "counts": c} for c in counts], as_counts=True)
        assert np.array_equal(dense.cpu().numpy(), np.stack(masks)), "RLE expansion"
        # non-binary values (the producer writes alpha 153) count as set
        packed, bbox = ops.erode(np.stack(masks) * 153)
        exp = np.stack([oracle.erode3x3(m) for m in masks])
        assert np.array_equal(ops.unpack_bits(packed, W, bbox), exp), f"erode_pack {W}x{H}"
        packed2, bbox2 = ops.erode_rle(counts, W, H)
        assert np.array_equal(ops.unpack_bits(packed2, W, bbox2), exp), f"rle_erode_pack {W}x{H}"
        for i, e in enumerate(exp):
            ys, xs = np.nonzero(e)
            want = [xs.min(), ys.min(), xs.max(), ys.max()] if xs.size else [0x7FFFFFFF, 0x7FFFFFFF, -1, -1]
            assert bbox.cpu().numpy()[i, :4].tolist() == want and bbox2.cpu().numpy()[i, :4].tolist() == want


def test_lane_nn_grid_equals_brute_force(oracle):
    """The grid/ring search must return exactly what the float64 brute force (the oracle, pinned to the
    reference by G6) returns: table sizes from 1 point to 50k, far-away centroids, centroids on lane
    points, duplicates (first index wins), several tables in one call."""
    from cm3d_amd import ops, synthetic as syn
    rng = np.random.default_rng(0)
    for trial in range(10):
        L = int([1, 7, 300, 5000, 50000][trial % 5])
        lane = syn.make_lane_table([600.0, 1600.0], L, seed=trial, extent=float([50, 200, 450][trial % 3]))
        if L > 3:
            lane[L // 2] = lane[L // 3]
        K = 300
        cent = np.stack([600 + rng.uniform(-500, 500, K), 1600 + rng.uniform(-500, 500, K), np.zeros(K)], 1).astype(np.float32)
        lane32 = lane.astype(np.float32)
        cent[:20, :2] = lane32[rng.integers(0, L, 20), :2]
        cent[20:40, :2] = ((lane32[rng.integers(0, L, 20), :2].astype(np.float64) + lane32[rng.integers(0, L, 20), :2]) / 2).astype(np.float32)
        yaws, dists, coords = ops.lane_yaws_distances_and_coords(cent, lane)
        j, d = oracle.lane_nn(cent, lane)
        assert np.array_equal(dists, d), f"trial {trial} L={L}: {(dists != d).sum()} distances differ"
        assert np.array_equal(yaws, lane32[j, 2]) and np.array_equal(coords, lane32[j, :2]), f"trial {trial}"


def test_mfma_distances_equal_the_vector_fma_chain():
    """The first pass over long medoid lists (k_medoid_approx) takes torch.cdist's five-term float32 fma chain from three
    v_mfma_f32_32x32x2_f32: on 2048 x 200 tiles of 32 x 32 point pairs at global-frame magnitudes (clusters 3 m and 40 m wide)
    every value must equal the vector pipe's chain bit for bit."""
    import torch
    from cm3d_amd import _lib
    L = _lib.lib()
    n_bad = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for seed in (1, 0xC0FFEE):
        _lib.check(L.cm3d_selftest_mfma(seed, 200, n_bad.data_ptr(), torch.cuda.current_stream().cuda_stream), "cm3d_selftest_mfma")
        torch.cuda.synchronize()
        assert int(n_bad.item()) == 0


def test_medoid_sqrt_is_correctly_rounded_on_its_whole_domain():
    """The medoid kernel's packed square root (rsq + coupled Newton step + residual correction), its neighbour-test
    reference form and sqrtf() agree bit for bit on EVERY float32 in [1e-30, 1e30) -- 1.67e9 values, checked on
    the device through the C-ABI diagnostic."""
    import torch
    from cm3d_amd import _lib
    L = _lib.lib()
    lo = int(np.float32(1e-30).view(np.uint32)) - 8
    hi = int(np.float32(1e30).view(np.uint32)) + 8
    n_bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    first = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(L.cm3d_selftest_sqrt(lo, hi, n_bad.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "cm3d_selftest_sqrt")
    torch.cuda.synchronize()
    assert hi - lo > 1.6e9
    assert int(n_bad.item()) == 0, f"{int(n_bad.item())} mismatches, first at bits 0x{int(first.item()) & 0xFFFFFFFF:08x}"


def test_integration_md_binding_runs_as_written(oracle):
    """The ctypes stub printed in INTEGRATION.md (what a maintainer of the reference would paste next to 2d_to_3d.py) is
    executed verbatim -- only the library path is filled in -- and must give the oracle's index lists."""
    import os
    import re
    import torch
    from cm3d_amd import _lib, synthetic as syn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libcm3d_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg = syn.config("tiny")
    fr = syn.make_frame(cfg, 11)
    pts = np.concatenate([oracle.sweep_prep(r, x[0:9], x[9:12], x[12:21], x[21:24]) for r, x in zip(fr.sweeps_raw, fr.sweep_xf)], 0)
    masks = np.stack([oracle.rle_decode(r).T for r in fr.rles])                     # (n,H,W) like depth_images (:425-428)
    want, _, _ = oracle.lift_frame_reference_order(pts, fr.cams, list(masks), fr.cam_nums)
    got = ns["points_in_all_masks"](torch.from_numpy(np.ascontiguousarray(pts.T)).cuda(), torch.from_numpy(fr.cams).cuda(),
                                    torch.from_numpy(np.ascontiguousarray(masks * 153)).cuda(), fr.cam_nums)
    assert len(got) == len(want) and sum(w.size for w in want) > 50
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


def test_projection_division_shortcut_equals_ieee_division():
    """The projection kernel's division sequence (no v_div_scale / v_div_fixup, shared refined reciprocal) against the
    IEEE division on 4e9 pseudo-random pairs over its domain, quotients next to integers over-represented."""
    import torch
    from cm3d_amd import _lib
    L = _lib.lib()
    n_bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    for seed in (1, 2):
        _lib.check(L.cm3d_selftest_div(seed, 2_000_000_000, n_bad.data_ptr(), torch.cuda.current_stream().cuda_stream), "cm3d_selftest_div")
        torch.cuda.synchronize()
        bad, benign = (int(v) for v in n_bad.cpu())
        # [1]: numerators below 2^-103, both quotients below 1 in magnitude -- never an accepted pixel
        assert bad == 0, f"{bad} quotients of magnitude >= 1/8 differ (seed {seed}); {benign} tiny ones"


def test_two_pass_medoid_on_near_ties(oracle):
    """Long lists whose column sums are nearly or exactly tied -- the cases where the first (approximate) pass of the
    two-pass medoid cannot decide and many columns must be settled exactly: points on a circle and on a sphere at
    global-frame magnitudes, a lattice with many duplicated points, two identical far-apart clusters.  The position must
    be the one the exact one-pass route (want_colsum) and the oracle find."""
    from cm3d_amd import ops
    rng = np.random.default_rng(41)
    centre = np.array([612.0, 1634.0, 1.5])
    cases = {}
    a = np.linspace(0, 2 * np.pi, 700, endpoint=False)
    cases["circle"] = centre + np.stack([4 * np.cos(a), 4 * np.sin(a), 0 * a], 1)
    v = rng.normal(size=(1500, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    cases["sphere"] = centre + 3.0 * v
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(3), indexing="ij"), -1).reshape(-1, 3) * 0.25
    cases["lattice_dups"] = centre + g[rng.integers(0, g.shape[0], 1200)]
    c1 = rng.normal(0, 0.5, (400, 3))
    cases["twin_clusters"] = centre + np.concatenate([c1, c1 + [30.0, 0, 0]])
    cases["cloud_5000"] = centre + rng.normal(0, [2.0, 1.0, 0.5], (5000, 3))
    for name, pts in cases.items():
        p32 = pts.astype(np.float32)
        want = oracle.medoid(np.concatenate([p32, np.zeros((len(p32), 1), np.float32)], 1), np.arange(len(p32)))
        exact, _ = ops.get_medoid(p32.T, want_colsum=True)
        got = ops.get_medoid(p32.T)
        assert exact == want, name
        assert got == want, name


def test_first_pass_forms_of_the_long_lists(oracle):
    """The matrix-pipe first pass of lists longer than 512 points runs on coordinates scaled by 2^-16 (root and clamp_min_ in one
    instruction, csrc/medoid.hip md_approx_tile) wherever the squared norms of a chunk's rows and a tile's columns lie in
    {0} or [1e-10, 1e9), and unscaled elsewhere.  Lists on both sides of every bound, and lists that mix the forms chunk by chunk
    (256 rows) and tile by tile (64 columns): the position must be the one the exact one-pass route and the oracle find."""
    from cm3d_amd import ops
    rng = np.random.default_rng(5)

    def cloud(centre, n, spread=(1.5, 0.8, 0.4)):
        return np.asarray(centre, np.float64) + rng.normal(0, spread, (n, 3))
    cases = {
        "vehicle frame (norms of a few m^2)": cloud([8.0, -3.0, 0.5], 900),
        "1.7 km": cloud([612.0, 1634.0, 1.5], 1400),
        "31 km: norms just below 1e9": cloud([22000.0, 22300.0, 1.5], 800),
        "32 km: norms just above 1e9, unscaled": cloud([22400.0, 22400.0, 1.5], 800),
        "60 km": cloud([42000.0, 43000.0, 2.0], 700),
        "the origin and points within a micrometre of it among ordinary ones": np.concatenate(
            [cloud([8.0, -3.0, 0.5], 300), np.zeros((2, 3)), rng.normal(0, 3e-7, (5, 3)), cloud([8.0, -3.0, 0.5], 500)]),
        "a far chunk of rows behind near ones": np.concatenate([cloud([612.0, 1634.0, 1.5], 600), cloud([30000.0, 30000.0, 1.5], 300)]),
    }
    for name, pts in cases.items():
        p32 = np.ascontiguousarray(pts.astype(np.float32))
        assert p32.shape[0] > 512
        want = oracle.medoid(np.concatenate([p32, np.zeros((len(p32), 1), np.float32)], 1), np.arange(len(p32)))
        exact, _ = ops.get_medoid(p32.T, want_colsum=True)
        got = ops.get_medoid(p32.T)
        assert exact == want, name
        assert got == want, name


def test_second_pass_candidate_counts_and_batches_that_mix_the_routes(oracle):
    """r04's second pass (csrc/medoid.hip k_medoid_long) settles up to 16 candidates per walk over the rows, further groups of 16 up to
    64, and more than 64 in the column loop; and a list of 257-448 points takes the two-pass route only in a batch that holds a list of
    more than 448.  Lists whose best point is duplicated K times (K tied candidates, the first of them must win) on both sides of every one of
    those limits, batches with and without a long list, both pointer forms of the entry point: every position must be the oracle's."""
    from cm3d_amd import ops
    rng = np.random.default_rng(11)
    centre = np.array([612.0, 1634.0, 1.5])

    def tied(M, K):
        pts = centre + rng.normal(0, [1.2, 0.7, 0.3], (M, 3))
        best = int(np.argmin(np.linalg.norm(pts - pts.mean(0), axis=1)))
        where = rng.choice(np.setdiff1d(np.arange(M), [best]), K - 1, replace=False) if K > 1 else []
        pts[where] = pts[best]
        return np.ascontiguousarray(pts.astype(np.float32))

    def want(p):
        return oracle.medoid(np.concatenate([p, np.zeros((len(p), 1), np.float32)], 1), np.arange(len(p)))
    lists = [tied(M, K) for M in (460, 700, 1500) for K in (1, 2, 15, 16, 17, 33, 64, 65, 130)]
    lists += [tied(M, K) for M in (257, 300, 448) for K in (1, 17, 70)] + [tied(100, 3), tied(256, 5), tied(64, 1)]
    exp = [want(p) for p in lists]
    for via_rows in (False, True):
        assert ops.get_medoids(lists, via_rows=via_rows) == exp, via_rows                           # one batch: every list beyond 256 in two passes
        short = [p for p in lists if len(p) <= 448]
        assert ops.get_medoids(short, via_rows=via_rows) == [want(p) for p in short], via_rows      # no list beyond 448: one pass for all
    one, _ = ops.get_medoids(lists, want_colsum=True)                                                # the one-pass route for every length
    assert one == exp


def test_every_route_of_the_medoid_root_equals_the_oracle(oracle):
    """k_medoid_tiles takes the root of a step of squared distances on one of four routes (csrc/medoid.hip md_rows): without any
    test when the norms of the list prove every value to be 0 or inside [1e-30, 1e30) (SAFE), packed after a test of the
    step's extremes, packed with zeros among the values, or sqrtf().  Lists built to land on each: ordinary global-frame
    points with duplicates and near neighbours (zeros everywhere, SAFE); the same list with the map origin itself and
    points within 1e-12 m of it among the columns (not SAFE: tested routes, zeros); coordinates of 1e-15 m (values around and
    below 1e-30: sqrtf); coordinates of 1e15 m (norms beyond 2e29: not SAFE, values beyond 1e30: sqrtf).  Column sums bit for
    bit the oracle's, on lists shorter and longer than one staging chunk."""
    from cm3d_amd import ops
    rng = np.random.default_rng(77)
    centre = np.array([612.0, 1634.0, 1.5])
    for M in (90, 700):
        base = centre + rng.normal(0, [0.6, 0.4, 0.3], (M, 3))
        base[rng.integers(0, M, M // 8)] = base[rng.integers(0, M, M // 8)]          # duplicated rows
        near_origin = base.copy()
        near_origin[5] = 0.0
        near_origin[6:12] = rng.normal(0, 1e-13, (6, 3))
        near_origin[70] = 0.0
        cases = {"global": base, "origin among the columns": near_origin,
                 "femtometres": rng.normal(0, 1e-15, (M, 3)), "femtometres and metres": np.concatenate([rng.normal(0, 1e-15, (M // 2, 3)), base[: M - M // 2] - centre]),
                 "1e15 m": rng.normal(0, 1e15, (M, 3)), "1e15 m among ordinary points": np.concatenate([base[: M - 3], rng.normal(0, 1e15, (3, 3))])}
        for name, pts in cases.items():
            p = np.ascontiguousarray(pts.astype(np.float32))
            j, cs = ops.get_medoid(p.T, want_colsum=True)
            P4 = np.concatenate([p, np.zeros((M, 1), np.float32)], 1)
            je, exp = oracle.medoid(P4, np.arange(M), want_colsum=True)
            assert np.array_equal(cs.view(np.uint32), exp.view(np.uint32)), (name, M, int((cs.view(np.uint32) != exp.view(np.uint32)).sum()))
            assert j == je, (name, M)


_RLE_FORM_SCRIPT = """
import sys
import numpy as np
sys.path.insert(0, {root!r})
from cm3d_amd import ops, rle
from oracle import oracle as orc
rng = np.random.default_rng(11)
for (W, H) in [(100, 33), (1024, 576), (1600, 900)]:
    masks = []
    for k in range(12):                      # ellipse-like blobs of very different sizes, some touching the border
        m = np.zeros((H, W), np.uint8)
        cx, cy = rng.uniform(-0.1, 1.1) * W, rng.uniform(-0.1, 1.1) * H
        ax, ay = rng.uniform(2, W / 3), rng.uniform(2, H / 2)
        yy, xx = np.mgrid[0:H, 0:W]
        m[((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0] = 1
        masks.append(m)
    masks.append(np.ones((H, W), np.uint8)); masks.append(np.zeros((H, W), np.uint8))
    masks.append((rng.random((H, W)) < 0.97).astype(np.uint8))             # tens of thousands of runs
    stripes = np.zeros((H, W), np.uint8); stripes[:, (np.arange(W) % 7) < 4] = 1; stripes[H // 2:, :] ^= 1
    masks.append(stripes)
    counts = [rle.dense_to_counts(m) for m in masks]
    exp = np.stack([orc.erode3x3(m) for m in masks])
    packed, bbox = ops.erode_rle(counts, W, H)
    assert np.array_equal(ops.unpack_bits(packed, W, bbox), exp), (W, H)
    for i, e in enumerate(exp):
        ys, xs = np.nonzero(e)
        want = [xs.min(), ys.min(), xs.max(), ys.max()] if xs.size else [0x7FFFFFFF, 0x7FFFFFFF, -1, -1]
        assert bbox.cpu().numpy()[i, :4].tolist() == want, (W, H, i)
print("FORM OK")
"""


@pytest.mark.parametrize("env", [{"CM3D_RLE_FORM": "wave"}, {"CM3D_RLE_FORM": "block"}, {"CM3D_RLE_FORM": "wave", "CM3D_RLE_BANDS": "4"},
                                 {"CM3D_RLE_FORM": "wave", "CM3D_RLEW_LDS_WORDS": "512"}])
def test_every_form_of_the_rle_kernel_equals_the_oracle(env):
    """cm3d_rle_erode_pack picks between a wave per mask and a workgroup per mask by the batch's average run count; here each form
    (and the row-band variant, and a small LDS tile that forces several tiles per mask) is FORCED over the same masks -- blobs from
    a few pixels to half the image, border-touching, full, empty, 30 000-run noise, stripes -- against the oracle's erosion."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RLE_FORM_SCRIPT.format(root=root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FORM OK" in r.stdout, r.stderr[-3000:]


_CP_SPAN_SCRIPT = """
import sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
from cm3d_amd import lifting, synthetic as syn
from oracle import oracle as orc
from tests.helpers import oracle_batch
from tests.test_gpu_parity import _compare
for name, over, n in (("tiny", dict(), 5), ("c1", dict(n_masks=24), 2)):
    cfg = syn.config(name, **over)
    frames = [syn.make_frame(cfg, 70 + i) for i in range(n)]
    lanes = [syn.make_lane_table(frames[0].ego_xyz[:2], 3000, seed=1)]
    hb = lifting.pack_frames(frames, lanes, [0] * n)
    exp = oracle_batch(orc, frames, lanes, [0] * n, hb)
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    eng.check_status()
    _compare(hb, eng.download(), exp)
print("SPAN OK")
"""


@pytest.mark.parametrize("span", ["2", "8"])
def test_every_span_of_the_compaction_equals_the_oracle(span):
    """cm3d_compact_hits gives a wave 4 wave-chunks (8 for frames of 150 k points and more); CM3D_CP_SPAN forces 2 / 4 / 8 (read once per
    process, hence the child): the same index lists, coordinates and everything behind them under each."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CP_SPAN_SCRIPT.format(root=root)], env=dict(os.environ, CM3D_CP_SPAN=span), capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "SPAN OK" in r.stdout, r.stderr[-3000:]


def test_g2e_sweep_loop_of_the_reference():
    """G2e on the device: raw sweeps -> fused sweep preparation + projection + compaction against what the REFERENCE'S OWN sweep
    loop (from_file, ego-box filter, rotate / translate twice, hstack; gen_golden_chain.py) and its loop body produced -- the
    cloud bit for bit (sha256), the dropped rows, the index lists and the listed points' coordinates, at 0 m / 1.7 km / 4 km."""
    import hashlib
    import torch
    from cm3d_amd import lifting, synthetic as syn
    from tests.test_oracle_golden import _g2e_cases
    for c in _g2e_cases():
        f = c["f"]
        hb = lifting.pack_frames([f], [syn.make_lane_table(f.ego_xyz[:2], 2000, seed=1)], [0])
        eng = lifting.LiftEngine(keep_cloud=True)
        eng.upload(hb)
        eng.run(masks="rle")
        torch.cuda.synchronize()
        got = eng.download()
        assert got["points"].shape[0] == c["n"] and hashlib.sha256(np.ascontiguousarray(got["points"]).tobytes()).hexdigest() == c["sha"], c["mag"]
        # the rows the device marked as dropped = the rows the reference's filter removed (sweep by sweep, in frame-local numbering)
        removed = eng.removed_rows()
        first = np.concatenate([[0], np.cumsum([np.asarray(r).shape[0] for r in f.sweeps_raw])])
        want = np.concatenate([c["dropped"][c["dropped_off"][s]:c["dropped_off"][s + 1]] + first[s] for s in range(len(f.sweeps_raw))])
        assert np.array_equal(np.flatnonzero(removed[:first[-1]]), want), c["mag"]
        assert np.array_equal(got["hit_off"], c["idx_off"]) and np.array_equal(got["hit_idx"], c["idx"]), c["mag"]
        assert np.array_equal(got["hit_xyz"].view(np.uint32), got["points"][c["idx"]].view(np.uint32))


def test_g7r_scene_chained_through_the_reference_functions(tmp_path):
    """G7r on the device: the scene the reference's own functions were chained over (stage 1, lane search, priors, push_centroid,
    per-sample NMS with the reference's driver loops; gen_golden_chain.py) through the lifting engine: the same boxes in the same
    order, centre and rotation within 1e-4 (north_star), here ~1e-6 (the device's float32 cos / sin of the lane yaw)."""
    import json
    import torch
    from cm3d_amd import lifting
    from tests.helpers import g7r_scene
    frames, lane = g7r_scene(tmp_path)
    want = json.load(open(os.path.join(G, "g7r_tiny_scene.json")))["results"]
    hb = lifting.pack_frames(frames, [lane], [0] * len(frames))
    eng = lifting.LiftEngine()
    eng.upload(hb)
    eng.run(masks="rle")
    torch.cuda.synchronize()
    got = lifting.box_records(hb, eng.download(full=False))
    assert set(got) == set(want)
    worst = 0.0
    for tok, boxes in want.items():
        assert [(b["detection_name"], b["detection_score"], b["size"], b["attribute_name"]) for b in got[tok]] == \
               [(b["detection_name"], b["detection_score"], b["size"], b["attribute_name"]) for b in boxes], tok
        for a, b in zip(got[tok], boxes):
            worst = max(worst, float(np.abs(np.array(a["translation"]) - np.array(b["translation"])).max()),
                        float(np.abs(np.array(a["rotation"]) - np.array(b["rotation"])).max()))
    assert sum(len(v) for v in want.values()) == 12 and worst < 1e-4
