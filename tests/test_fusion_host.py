"""CPU: the oracle's restatement of the SAM3D fusion matching (reference
src/nuscenes/linear_matching.py:53-121,231-259) against known answers and independent solvers, and the host logic
of cm3d_amd.fusion (:142-491).  The matcher's C++ (waymo_open_dataset py_metrics_ops.match) is not in the reference
checkout: parity unpinned; what is pinned here is geometry (analytic areas, rasterisation) and optimality."""
import itertools

import numpy as np
import pytest


def _rand_boxes(rng, n, centre=(0.0, 0.0), spread=6.0):
    c = np.asarray(centre) + rng.uniform(-spread, spread, (n, 2))
    return np.stack([c[:, 0], c[:, 1], rng.uniform(-1, 1, n), rng.uniform(1.5, 5.5, n), rng.uniform(0.8, 2.5, n),
                     rng.uniform(1, 2, n), rng.uniform(-np.pi, np.pi, n)], axis=1)


def _records(b7):
    b = np.asarray(b7, np.float64).reshape(-1, 7).astype(np.float32).astype(np.float64)
    return np.stack([b[:, 0], b[:, 1], b[:, 3], b[:, 4], np.cos(b[:, 6]), np.sin(b[:, 6])], axis=1)


def test_iou_known_answers(oracle):
    B = oracle.bev_box
    sq = B(0, 0, 2, 2, 0)
    assert oracle.bev_iou(sq, sq) == 1.0
    assert oracle.bev_iou(sq, B(1, 0, 2, 2, 0)) == pytest.approx(1 / 3, abs=1e-15)
    oct_area = 8 * (np.sqrt(2) - 1)                       # square against itself turned by 45 degrees
    assert oracle.bev_iou(sq, B(0, 0, 2, 2, np.pi / 4)) == pytest.approx(oct_area / (8 - oct_area), abs=1e-14)
    assert oracle.bev_iou(sq, B(5, 0, 2, 2, 0.3)) == 0.0                      # disjoint
    assert oracle.bev_iou(sq, B(2.0, 0, 2, 2, 0)) == 0.0                      # sharing an edge
    assert oracle.bev_iou(B(0, 0, 4, 4, 0.7), B(0.2, -0.1, 1, 1, 0.1)) == pytest.approx(1 / 16, abs=1e-14)    # contained
    assert oracle.bev_iou(sq, B(0, 0, 2, 2, np.pi / 2)) == pytest.approx(1.0, abs=1e-14)
    assert oracle.bev_iou(B(0, 0, 4, 2, 0), B(0, 0, 4, 2, np.pi / 2)) == pytest.approx(4 / 12, abs=1e-14)
    assert oracle.bev_iou(sq, B(0, 0, 0, 0, 0)) == 0.0 and oracle.bev_iou(B(0, 0, 0, 0, 0), sq) == 0.0       # zeros(D): no box
    # global-frame magnitudes: same overlap as at the origin
    a, b = B(0.3, -0.2, 4.5, 1.9, 0.4), B(1.0, 0.4, 4.2, 2.0, 0.9)
    a2, b2 = a.copy(), b.copy()
    a2[:2] += (640.0, 1620.0); b2[:2] += (640.0, 1620.0)
    assert oracle.bev_iou(a, b) == pytest.approx(oracle.bev_iou(a2, b2), abs=1e-12)


def test_iou_symmetric_and_equal_to_rasterisation(oracle):
    rng = np.random.default_rng(11)
    xs = (np.arange(1200) + 0.5) / 1200 * 24 - 12
    X, Y = np.meshgrid(xs, xs)

    def inside(r):
        dx, dy = X - r[0], Y - r[1]
        u, v = dx * r[4] + dy * r[5], -dx * r[5] + dy * r[4]
        return (np.abs(u) <= r[2] / 2) & (np.abs(v) <= r[3] / 2)
    recs = _records(_rand_boxes(rng, 24, spread=1.5))
    n_pos = 0
    for i in range(0, 24, 2):
        a, b = recs[i], recs[i + 1]
        iou = oracle.bev_iou(a, b)
        assert iou == pytest.approx(oracle.bev_iou(b, a), abs=1e-13)
        ia, ib = inside(a), inside(b)
        ref = (ia & ib).sum() / max((ia | ib).sum(), 1)
        assert abs(iou - ref) < 4e-3
        n_pos += iou > 0
    assert n_pos >= 8


def _brute_force(W):
    P, G = W.shape
    best = 0
    if P <= G:
        for cols in itertools.permutations(range(G), P):
            best = max(best, sum(int(W[i, c]) for i, c in enumerate(cols)))
    else:
        for rows in itertools.permutations(range(P), G):
            best = max(best, sum(int(W[r, j]) for j, r in enumerate(rows)))
    return best


def test_match_is_a_maximum_weight_assignment(oracle):
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    for P, G in [(1, 1), (1, 5), (5, 1), (3, 3), (4, 6), (6, 4), (5, 5), (6, 6), (2, 7)]:
        for rep in range(4):
            pred, gt = _records(_rand_boxes(rng, P, spread=2.5)), _records(_rand_boxes(rng, G, spread=2.5))
            pm, gm, iou, total, W = oracle.bev_match(pred, gt, 0.2, want_weights=True)
            assert total == _brute_force(W), (P, G, rep)
            ids = np.flatnonzero(pm >= 0)
            assert len(set(pm[ids])) == len(ids)                                   # one to one
            assert all(gm[pm[i]] == i for i in ids) and (gm >= 0).sum() == len(ids)
            assert all(W[i, pm[i]] >= 200000 and iou[i] >= 0.2 for i in ids)       # nothing below the threshold survives
            assert all(iou[i] == 0.0 for i in range(P) if pm[i] < 0)
    for P, G in [(40, 60), (60, 40), (100, 100)]:
        pred, gt = _records(_rand_boxes(rng, P, spread=12)), _records(_rand_boxes(rng, G, spread=12))
        pm, gm, iou, total, W = oracle.bev_match(pred, gt, 0.2, want_weights=True)
        r, c = linear_sum_assignment(W, maximize=True)
        assert total == int(W[r, c].sum()) and total > 0


def test_match_threshold_and_empty_sides(oracle):
    B = oracle.bev_box
    pred = np.stack([B(0, 0, 4, 2, 0), B(10, 0, 4, 2, 0)])
    gt = np.stack([B(10.5, 0, 4, 2, 0), B(3.0, 0, 4, 2, 0), B(50, 50, 4, 2, 0)])
    pm, gm, iou, total = oracle.bev_match(pred, gt, 0.2)
    assert pm.tolist() == [-1, 0] and gm.tolist() == [1, -1, -1]          # IoU(pred0, gt1) = 1/7 < 0.2
    assert iou[1] == pytest.approx(3.5 / 4.5, abs=1e-14) and total == int(3.5 / 4.5 * 1e6)
    pm, gm, iou, total = oracle.bev_match(pred, gt, 0.1)
    assert pm.tolist() == [1, 0]
    pm, gm, iou, total = oracle.bev_match(pred, np.zeros((0, 6)), 0.2)
    assert pm.tolist() == [-1, -1] and gm.size == 0 and total == 0


# ------------------------------------------------------------------ host logic of cm3d_amd.fusion
def _obj(tok, xyz, size, yaw, name, score):
    return {"sample_token": tok, "translation": list(xyz), "size": list(size), "rotation": [float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2))],
            "velocity": [0, 0], "detection_name": name, "detection_score": score, "attribute_name": ""}


def test_heading_quirk_and_quaternion():
    from cm3d_amd import fusion, geometry as geo
    for yaw in (-2.5, -0.3, 0.0, 0.7, 3.0):
        h = fusion.heading_of([np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)])
        assert np.isclose(np.angle(np.exp(1j * (h - (np.pi - yaw)))), 0.0, atol=1e-12)          # :171 reads wxyz as xyzw
    for h in (-3.0, -1.6, -0.2, 0.0, 1.0, 1.58, 3.1):
        q = fusion.yaw_quaternion(h)
        assert np.allclose(geo.quat_to_rotmat(q), geo.rot_z(h), atol=1e-12) and q[1] == q[2] == 0.0
        assert (q[3] > 0) if np.cos(h) < 0 else (q[0] > 0)                                      # pyquaternion's trace-method branches


def test_alpha_grid_and_score_range():
    from cm3d_amd import fusion
    res = {"s": [_obj("s", (0, 0, 1), (2, 4, 1.5), 0, "car", 0.0), _obj("s", (5, 0, 1), (2, 4, 1.5), 0, "car", 0.4),
                 _obj("s", (9, 0, 1), (2, 4, 1.5), 0, "car", 0.8)]}
    _, _, mx, mn = fusion.parse_results(res, zero_min_quirk=True)
    assert (mx, mn) == (0.8, 0.4)                                                               # :186-190
    _, _, mx, mn = fusion.parse_results(res)
    assert (mx, mn) == (0.8, 0.0)
    a = fusion.alpha_grid(0.1, 0.9, 0.4, 0.8)
    assert a[0] == 0.1 / 0.8 and np.allclose(np.diff(a), 0.04) and a[-1] < 0.9 / 0.4 <= a[-1] + 0.04 + 1e-12


def test_fuse_groups_and_scores(oracle, monkeypatch):
    """fuse() on hand-made files; the GPU matcher is replaced by the oracle for this CPU test."""
    from cm3d_amd import fusion, ops

    def cpu_match(pred_boxes, gt_boxes, iou=0.2):
        out = []
        for p, g in zip(pred_boxes, gt_boxes):
            pm, gm, io, _ = oracle.bev_match(ops.match_records(p), ops.match_records(g), iou)
            ids = np.flatnonzero(pm >= 0)
            out.append((ids, pm[ids], io[ids]))
        return out
    monkeypatch.setattr(ops, "bev_match", cpu_match)
    pred = {"results": {
        "a": [_obj("a", (0, 0, 1), (4, 2, 1.5), 0.0, "car", 0.5), _obj("a", (20, 0, 1), (4, 2, 1.5), 0.0, "truck", 0.6)],
        "b": [_obj("b", (0, 0, 1), (4, 2, 1.5), 0.0, "bus", 0.3)],
        "c": []}}
    sam = {"results": {
        "a": [_obj("a", (40, 0, 1), (4, 2, 1.5), 0.0, "object", 0.9), _obj("a", (0.4, 0.1, 1.1), (4.2, 2.1, 1.6), 0.05, "object", 0.8)],
        "d": [_obj("d", (1, 1, 1), (1, 1, 1), 0.0, "object", 0.7)]}}
    pb, ps, pmax, pmin = fusion.parse_results(pred["results"])
    sb, ss, smax, smin = fusion.parse_results(sam["results"], zero_min_quirk=True)
    pm, sm = fusion.match_samples(pb, sb, 0.2)
    assert pm == {"a": [0], "b": [], "c": []} and sm == {"a": [1], "b": [], "c": []}
    # alpha small: the matched pair keeps the prediction's box and score
    fused, n = fusion.fuse(pb, ps, sb, ss, pm, sm, 0.5)
    assert n == dict(num_samples=3, num_pred_boxes=2, num_sam3d_boxes=2, num_sam3d_samples=2, num_matched_boxes=1)
    A = fused["results"]["a"]
    assert [b["detection_name"] for b in A] == ["truck", "object", "car"]                      # unmatched pred, unmatched sam3d, matched
    assert A[1]["detection_score"] == 0.45 and A[2]["detection_score"] == 0.5
    assert np.allclose(A[2]["translation"], [0, 0, 1]) and A[2]["size"] == [4.0, 2.0, 1.5]
    assert list(fused["results"]) == ["a", "b", "d"] and fused["meta"]["use_lidar"] is True
    # alpha large: the SAM3D box wins, keeps the prediction's name, score clipped to 1
    fused, _ = fusion.fuse(pb, ps, sb, ss, pm, sm, 2.0)
    m = fused["results"]["a"][2]
    assert m["detection_name"] == "car" and m["detection_score"] == 1.0 and m["size"] == [4.2, 2.1, 1.6]
    assert np.allclose(m["translation"], [0.4, 0.1, 1.1])
    # headings go through the wxyz-as-xyzw reading: yaw 0.05 comes back as pi - 0.05
    assert np.allclose(m["rotation"], fusion.yaw_quaternion(np.pi - 0.05), atol=1e-12)


def test_waymo_fusion_host_logic(oracle, monkeypatch, tmp_path):
    """src/waymo/linear_matching.py on decoded Objects: grid, three groups, metrics parser; matcher = oracle on the CPU."""
    from cm3d_amd import fusion, ops, waymo as wm

    def cpu_match(pred_boxes, gt_boxes, iou=0.2):
        out = []
        for p, g in zip(pred_boxes, gt_boxes):
            pm, gm, io, _ = oracle.bev_match(ops.match_records(p), ops.match_records(g), iou)
            ids = np.flatnonzero(pm >= 0)
            out.append((ids, pm[ids], io[ids]))
        return out
    monkeypatch.setattr(ops, "bev_match", cpu_match)

    def obj(ctx, ts, c, lwh, heading, typ, score, oid):
        return wm.encode_object(c, lwh[0], lwh[1], lwh[2], heading, typ, score, ctx, ts, object_id=oid)
    pred = wm.decode_objects(wm.encode_objects([obj("seg", 10, [0, 0, 1], [4, 2, 1.5], 0.1, 1, 0.5, "p0"),
                                                obj("seg", 10, [30, 0, 1], [4, 2, 1.5], 0.0, 1, 0.25, "p1"),
                                                obj("seg", 20, [5, 5, 1], [1, 1, 1.8], 0.0, 2, 0.75, "p2")]))
    sam = wm.decode_objects(wm.encode_objects([obj("seg", 10, [0.3, 0.1, 1.1], [4.2, 2.1, 1.6], 0.15, 0, 0.5, "s0"),
                                               obj("seg", 10, [60, 0, 1], [4, 2, 1.5], 0.0, 0, 0.25, "s1"),
                                               obj("seg", 30, [1, 1, 1], [2, 2, 2], 0.0, 0, 0.0, "s2")]))
    pb, ps, pmax, pmin = fusion.waymo_parse(pred)
    sb, ss, smax, smin = fusion.waymo_parse(sam, zero_min_quirk=True)
    assert (pmax, pmin, smax, smin) == (0.75, 0.25, 0.5, 0.25)
    grid = fusion.waymo_alpha_grid(pmin, pmax, smin, smax)
    full = np.arange(0.5, 3.0 + 0.04, 0.04)
    assert np.allclose(grid, full[::-1][3:]) and grid[0] > grid[-1]
    pm, sm = fusion.match_samples(pb, sb, 0.2)
    assert pm == {("seg", 10): [0], ("seg", 20): []} and sm == {("seg", 10): [0], ("seg", 20): []}
    got = wm.decode_objects(wm.encode_objects(fusion.fuse_waymo(pb, ps, sb, ss, pm, sm, 2.0)))
    assert [o["id"] for o in got] == ["p1", "p2", "s1", "s2", "p0"]
    m = got[-1]                                     # SAM3D box wins (0.5 * 2 > 0.5), keeps the prediction's id and type, score clipped
    assert m["type"] == 1 and m["score"] == 1.0 and m["length"] == np.float64(4.2) and np.allclose(m["center"], [0.3, 0.1, 1.1])
    assert got[2]["score"] == 0.5 and got[3]["score"] == 0.0
    got = wm.decode_objects(wm.encode_objects(fusion.fuse_waymo(pb, ps, sb, ss, pm, sm, 0.5)))
    assert got[-1]["score"] == 0.5 and got[-1]["length"] == 4.0 and np.allclose(got[-1]["center"], [0, 0, 1])
    # metrics text of compute_detection_metrics_main
    names = ["VEHICLE", "PEDESTRIAN", "SIGN", "CYCLIST"]
    text = "".join(f"OBJECT_TYPE_TYPE_{n}_LEVEL_{l}: [mAP {0.1 * (i + 1) + 0.01 * l}] [mAPH {0.05 * (i + 1) + 0.01 * l}]\n"
                   for i, n in enumerate(names) for l in (1, 2))
    ap, score = fusion.parse_waymo_metrics(text)
    assert ap["Vehicle/L1 mAP"] == 0.11 and ap["Cyclist/L2 mAPH"] == pytest.approx(0.22) and ap["Sign/L2 mAP"] == pytest.approx(0.32)
    assert score == pytest.approx((0.12 + 0.22 + 0.42) / 3)
    # grid search with a stand-in evaluator: best file = the alpha the evaluator prefers
    calls = []

    def evaluate(path):
        objs = wm.decode_objects(open(path, "rb").read())
        calls.append(len(objs))
        return -abs(objs[-1]["score"] - 0.9)
    a, sc = fusion.waymo_grid_search(pred, sam, evaluate, str(tmp_path / "m.bin"), str(tmp_path / "best.bin"), verbose=False)
    assert len(calls) == len(grid) and abs(a - 1.8) < 0.03
    best = wm.decode_objects(open(tmp_path / "best.bin", "rb").read())
    assert best[-1]["score"] == pytest.approx(0.5 * a)
