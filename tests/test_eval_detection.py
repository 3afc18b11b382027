"""CPU: the detection evaluation harness against hand-computed
known answers and on the synthetic dataset.  The reference module needs nuscenes-devkit at import time, so there
is no golden vector for it (parity unpinned, see the module docstring)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cm3d_amd import eval_detection as ev

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _boxes(per_sample):
    out = ev.EvalBoxes()
    for tok, rows in per_sample.items():
        out.add_boxes(tok, [ev._box(tok, r["t"], r.get("size", (2, 4, 1.5)), r.get("rot", (1, 0, 0, 0)), r.get("vel", (0, 0)), 5,
                                    r.get("name", "car"), r.get("score", -1.0), r.get("attr", "")) for r in rows])
    return out


def test_matching_precision_recall_and_ap_by_hand():
    gt = _boxes({"s": [{"t": (0, 0, 0)}, {"t": (10, 0, 0)}]})
    pred = _boxes({"s": [{"t": (0.3, 0, 0), "score": 0.9}, {"t": (5, 5, 0), "score": 0.8}, {"t": (10.2, 0, 0), "score": 0.7}]})
    md, rec_actual = ev.accumulate_object_class(gt, pred, None, 2.0)
    rec2, md2 = ev.accumulate_with_recall(gt, pred, "car", None, 2.0)
    assert rec_actual == 1.0 and rec2 == 1.0 and np.array_equal(md.precision, md2.precision)
    # tp = 1,0,1 -> precision 1, 1/2, 2/3 at recall 1/2, 1/2, 1: below recall 0.5 the curve is 1, at 0.5 it drops to 0.5
    # and rises linearly to 2/3 at recall 1
    r = np.linspace(0, 1, 101)
    want = np.where(r < 0.5, 1.0, 0.5 + (r - 0.5) / 0.5 * (2.0 / 3.0 - 0.5))
    assert np.allclose(md.precision, want, atol=1e-12)
    ap = ev.calc_ap(md, 0.1, 0.1)
    assert abs(ap - np.mean(np.maximum(want[11:] - 0.1, 0.0)) / 0.9) < 1e-12
    # translation error of the matches: 0.3 then mean(0.3, 0.2); aligned with the confidence curve
    assert abs(md.trans_err[md.max_recall_ind] - 0.25) < 1e-9
    assert abs(ev.calc_tp(md, 0.1, "trans_err") - float(np.mean(md.trans_err[11:md.max_recall_ind + 1]))) < 1e-12
    # a tighter threshold loses the 0.3 m match but keeps the 0.2 m one
    md_t, rec_t = ev.accumulate_object_class(gt, pred, None, 0.25)
    assert rec_t == 0.5


def test_greedy_matching_takes_ground_truth_once_and_respects_classes():
    gt = _boxes({"a": [{"t": (0, 0, 0), "name": "car"}], "b": [{"t": (0, 0, 0), "name": "pedestrian"}]})
    pred = _boxes({"a": [{"t": (0.1, 0, 0), "score": 0.5, "name": "car"}, {"t": (0.0, 0, 0), "score": 0.9, "name": "car"}],
                   "b": [{"t": (0, 0.1, 0), "score": 0.7, "name": "car"}]})
    md, rec = ev.accumulate_object_class(gt, pred, None, 1.0)          # class-agnostic: both GT found, one duplicate is a FP
    assert rec == 1.0
    rec_car, _ = ev.accumulate_with_recall(gt, pred, "car", None, 1.0)
    rec_ped, md_ped = ev.accumulate_with_recall(gt, pred, "pedestrian", None, 1.0)
    assert rec_car == 1.0 and rec_ped == 0 and np.all(md_ped.precision == 0)
    # no ground truth at all
    md0, rec0 = ev.accumulate_object_class(_boxes({"a": []}), pred, None, 1.0)
    assert rec0 == 0 and np.all(md0.trans_err == 1)


def test_per_match_measures():
    c, s = np.cos(0.25), np.sin(0.25)          # yaw 0.5 as a quaternion about z
    a = ev._box("s", (0, 0, 0), (2, 4, 1), (1, 0, 0, 0), (1.0, 0.0), attribute_name="vehicle.moving")
    b = ev._box("s", (3, 4, 9), (1, 4, 2), (c, 0, 0, s), (0.0, 2.0), attribute_name="vehicle.parked")
    assert ev.center_distance(a, b) == 5.0 and abs(ev.velocity_l2(a, b) - np.sqrt(5)) < 1e-12
    assert abs(ev.scale_iou(a, b) - 4.0 / (8 + 8 - 4)) < 1e-12
    assert abs(ev.yaw_diff(a, b) - 0.5) < 1e-12
    flipped = ev._box("s", (0, 0, 0), (2, 4, 1), (np.cos((np.pi + 0.1) / 2), 0, 0, np.sin((np.pi + 0.1) / 2)))
    assert abs(ev.yaw_diff(a, flipped, period=np.pi) - 0.1) < 1e-9 and abs(ev.yaw_diff(a, flipped) - (np.pi - 0.1)) < 1e-9
    assert ev.attr_acc(a, b) == 0.0 and np.isnan(ev.attr_acc(ev._box("s", (0, 0, 0), (1, 1, 1), (1, 0, 0, 0)), b))
    assert np.allclose(ev.cummean(np.array([1.0, np.nan, 3.0])), [1.0, 1.0, 2.0]) and np.all(ev.cummean(np.array([np.nan])) == 1)
    assert ev.category_to_detection_name("human.pedestrian.child") == "pedestrian"
    assert ev.category_to_detection_name("human.pedestrian.child", rare=True) == "child"
    assert ev.category_to_detection_name("animal") is None


@pytest.fixture()
def synthetic_eval(tmp_path, oracle):
    from cm3d_amd import lifting, nusc_io, synthetic as syn
    from tests.helpers import oracle_results
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=1, frames_per_scene=3)
    tables = nusc_io.NuscTables("v1.0-synth", dataroot)
    scene = tables.scene_by_name(names[0])
    frames = nusc_io.frames_of_scene(tables, scene, mask_dir, n_sweeps=3, ratio=cfg.ratio)
    lanes = [nusc_io.load_lane_points(dataroot, tables.location(scene))]
    results = oracle_results(oracle, frames, lanes, [0] * len(frames))
    path = tmp_path / "pseudolabels.json"
    json.dump({"meta": {"use_lidar": False}, "results": results}, open(path, "w"))
    return tables, dataroot, str(path), tmp_path


def test_ground_truth_against_itself_is_perfect(synthetic_eval):
    tables, dataroot, _, tmp = synthetic_eval
    gt = ev.load_gt(tables)
    assert len(gt.all) > 10 and all(b["num_pts"] == 10 for b in gt.all)
    perfect = {tok: [dict(sample_token=tok, translation=b["translation"], size=b["size"], rotation=b["rotation"], velocity=[0, 0],
                          detection_name=b["detection_name"], detection_score=0.5 + 0.001 * i, attribute_name="")
                     for i, b in enumerate(boxes)] for tok, boxes in gt.boxes.items()}
    path = tmp / "perfect.json"
    json.dump({"meta": {}, "results": perfect}, open(path, "w"))
    for object_only in (True, False):
        de = ev.DetectionEval(tables, ev.config_factory(), str(path), None, str(tmp / "out"), False, object_only, verbose=False)
        summary = de.main()
        n_in_range = len(de.gt_boxes.all)
        assert 0 < n_in_range <= len(gt.all)
        aps = [v for name, v in summary["mean_dist_aps"].items() if any(b["detection_name"] == name or object_only for b in de.gt_boxes.all)]
        assert aps and all(abs(a - 1.0) < 1e-9 for a in aps)
        present = ["object"] if object_only else sorted({b["detection_name"] for b in de.gt_boxes.all})
        for name in present:          # (classes without ground truth keep the devkit's error of 1)
            assert summary["label_tp_errors"][name]["trans_err"] < 1e-9 and summary["label_tp_errors"][name]["scale_err"] < 1e-9


def test_pseudo_labels_on_the_synthetic_scene(synthetic_eval):
    tables, dataroot, result_path, tmp = synthetic_eval
    de = ev.DetectionEval(tables, ev.config_factory(), result_path, None, str(tmp / "metrics"), False, True, verbose=False)
    summary = de.main()
    # the tiny synthetic scene (3000-point sweeps, random extra masks, duplicates) is no accuracy benchmark: the
    # check is that real pseudo-labels flow through matching, AP and the TP metrics and find some of the objects
    assert 0.02 < summary["mean_dist_aps"]["object"] <= 1.0 and 0.0 < summary["mean_recall"] <= 1.0
    assert 0.0 <= summary["tp_errors"]["trans_err"] < 2.0
    assert 0.0 <= summary["nd_score"] <= 1.0 and os.path.exists(tmp / "metrics" / "metrics_summary.json")
    # the command-line entry point gives the same numbers
    r = subprocess.run([sys.executable, "eval_custom.py", result_path, "--output_dir", str(tmp / "cli"), "--dataroot", dataroot,
                        "--version", "v1.0-synth", "--object_only", "1", "--verbose", "0"],
                       cwd=os.path.join(ROOT, "src", "nuscenes"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    cli = json.load(open(tmp / "cli" / "metrics_summary.json"))
    assert abs(cli["mean_ap"] - summary["mean_ap"]) < 1e-12


def test_drivable_area_filter_known_answers(synthetic_eval):
    """eval_custom's drivable-area filter (:489-526): a box survives only if its centre lies within a drivable polygon of the
    first sample's map.  The synthetic map is a 160 m square around the first ego position with a 24 m square hole."""
    tables, dataroot, _, tmp = synthetic_eval
    scene = tables.scenes()[0]
    polys = ev.load_drivable_polygons(dataroot, tables.location(scene))
    assert len(polys) == 1 and polys[0][0].shape == (4, 2) and len(polys[0][1]) == 1
    (x0, y0), (x1, y1) = polys[0][0].min(0), polys[0][0].max(0)
    cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
    assert ev.point_in_polygons(polys, cx, cy) and ev.point_in_polygons(polys, x0 + 1e-6, cy)
    assert not ev.point_in_polygons(polys, cx + 30.0, cy + 30.0)                 # in the hole
    assert ev.point_in_polygons(polys, cx + 19.9, cy + 30.0) and ev.point_in_polygons(polys, cx + 44.1, cy + 30.0)
    assert not ev.point_in_polygons(polys, x1 + 0.1, cy) and not ev.point_in_polygons(polys, cx, y0 - 5.0)
    tok = tables.samples_of_scene(scene)[0]["token"]
    boxes = ev.EvalBoxes()
    mk = lambda x, y: ev._box(tok, (x, y, 0.0), (2, 4, 1.5), (1, 0, 0, 0), detection_name="car", detection_score=0.5)
    boxes.add_boxes(tok, [mk(cx + 5, cy - 3), mk(cx + 30, cy + 30), mk(cx + 10, cy + 10), mk(cx + 200, cy)])
    ev.add_center_dist(tables, boxes)
    far = {k: 1e9 for k in ev.CVPR_2019["class_range"]}
    kept = ev.filter_eval_boxes(tables, boxes, far, drivable_filtering=True)
    assert [tuple(b["translation"][:2]) for b in kept[tok]] == [(cx + 5, cy - 3), (cx + 10, cy + 10)]
    # without the flag nothing is dropped; with it, ground truth is filtered the same way inside DetectionEval
    boxes2 = ev.EvalBoxes()
    boxes2.add_boxes(tok, [mk(cx + 30, cy + 30)])
    ev.add_center_dist(tables, boxes2)
    assert len(ev.filter_eval_boxes(tables, boxes2, far, drivable_filtering=False)[tok]) == 1


def test_rare_class_configuration(synthetic_eval, tmp_path):
    """cfg/rare_config.json (12 classes: + child, stroller; min_recall = min_precision = 0): with more than 10 classes the
    ground truth is loaded through category_to_detection_name_rare (:204-232,:927-930)."""
    tables, dataroot, result_path, tmp = synthetic_eval
    cfg = ev.DetectionConfig.deserialize(json.load(open(os.path.join(ROOT, "src", "nuscenes", "cfg", "rare_config.json"))))
    assert len(cfg.class_range) == 12 and cfg.class_range["child"] == 40 and cfg.min_recall == 0 and cfg.min_precision == 0
    # turn one pedestrian annotation of the synthetic scene into a child: rare ground truth keeps it apart
    child_cat = "cat-human.pedestrian.child"
    tables.t["category"][child_cat] = {"token": child_cat, "name": "human.pedestrian.child"}
    ann = next(a for a in tables.t["sample_annotation"].values() if tables.category_name(a) == "human.pedestrian.adult")
    tables.t["instance"][ann["instance_token"]]["category_token"] = child_cat
    plain = ev.load_gt(tables, rare=False)
    rare = ev.load_gt(tables, rare=True)
    assert sum(b["detection_name"] == "child" for b in plain.all) == 0 and sum(b["detection_name"] == "child" for b in rare.all) == 1
    assert len(plain.all) == len(rare.all)
    de = ev.DetectionEval(tables, cfg, result_path, None, str(tmp_path / "rare"), False, False, verbose=False)
    assert any(b["detection_name"] == "child" for b in de.gt_boxes.all)
    summary = de.main()
    assert set(summary["mean_dist_aps"]) == set(cfg.class_range) and summary["mean_dist_aps"]["child"] == 0.0


# ---------------------------------------------------------------------------------------------------------------- G10
def _g10():
    import gzip
    with gzip.open(os.path.join(ROOT, "tests", "golden", "g10_eval.json.gz")) as f:
        return json.loads(f.read().decode())


def _g10_dataset(g, case, root):
    """The fixture's table set as a nuScenes directory (what NuscTables reads) + the map-expansion files of its drivable
    polygons + the case's predictions as a result file."""
    base = os.path.join(root, g["version"])
    os.makedirs(base, exist_ok=True)
    scenes = {s["token"] for s in g["tables"]["scene"] if s["name"] in case["scenes"]}
    samples = {s["token"] for s in g["tables"]["sample"] if s["scene_token"] in scenes}
    for name, rows in g["tables"].items():
        if name == "scene":
            rows = [r for r in rows if r["token"] in scenes]
        elif name == "sample":
            rows = [r for r in rows if r["token"] in samples]
        elif name in ("sample_data", "sample_annotation"):
            rows = [r for r in rows if r["sample_token"] in samples]
        with open(os.path.join(base, name + ".json"), "w") as f:
            json.dump(rows, f)
    os.makedirs(os.path.join(root, "maps", "expansion"), exist_ok=True)
    for loc, polys in g["polygons"].items():
        nodes, polygons = [], []
        for pi, (ext, holes) in enumerate(polys):
            def ring(pts, tag):
                toks = []
                for k, (x, y) in enumerate(pts):
                    toks.append(f"n-{pi}-{tag}-{k}")
                    nodes.append({"token": toks[-1], "x": x, "y": y})
                return toks
            polygons.append({"token": str(pi), "exterior_node_tokens": ring(ext, "e"),
                             "holes": [{"node_tokens": ring(h, f"h{hi}")} for hi, h in enumerate(holes)]})
        with open(os.path.join(root, "maps", "expansion", loc + ".json"), "w") as f:
            json.dump({"node": nodes, "polygon": polygons, "drivable_area": [{"token": "da0", "polygon_tokens": [p["token"] for p in polygons]}]}, f)
    res = os.path.join(root, f"results_{case['tag']}.json")
    with open(res, "w") as f:
        json.dump({"meta": {"use_lidar": False}, "results": {t: [dict(b, sample_token=t) for b in bs] for t, bs in case["predictions"].items()}}, f)
    return res


def _same_boxes(ours, theirs, what):
    assert list(ours.boxes.keys()) == list(theirs.keys()), what
    for tok, rows in theirs.items():
        got = ours[tok]
        assert len(got) == len(rows), (what, tok, len(got), len(rows))
        for a, b in zip(got, rows):
            assert a["detection_name"] == b["detection_name"] and a["attribute_name"] == b["attribute_name"] and a["num_pts"] == b["num_pts"], (what, tok)
            assert a["detection_score"] == b["detection_score"] and list(a["translation"]) == b["translation"] and list(a["size"]) == b["size"]
            assert list(a["rotation"]) == b["rotation"], (what, tok)
            va = [None if np.isnan(v) else v for v in a["velocity"]]
            assert len(va) == len(b["velocity"]) and all((x is None and y is None) or (x is not None and y is not None and abs(x - y) < 1e-12)
                                                        for x, y in zip(va, b["velocity"])), (what, tok, va, b["velocity"])
            assert np.allclose(a["ego_translation"], b["ego_translation"], rtol=0, atol=1e-12), (what, tok)


def _same_md(md, want, what):
    for k, v in want.items():
        got, exp = np.asarray(getattr(md, k), float), np.asarray(v, float)
        assert got.shape == exp.shape and np.allclose(got, exp, rtol=0, atol=1e-12, equal_nan=True), (what, k, float(np.nanmax(np.abs(got - exp))))


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_g10_harness_equals_the_references_own_functions(tmp_path, ci):
    """G10 (tests/golden/gen_golden_eval.py): load_gt, add_center_dist, filter_eval_boxes (incl. the drivable-area filter and
    the rare class mapping), accumulate_object_class and accumulate_with_recall of the REFERENCE's eval_custom.py, run unchanged
    on a synthetic table set (with restated third-party helpers underneath), against cm3d_amd.eval_detection on the same
    tables read from disk: the same boxes survive every stage in the same order, and every metric curve agrees to 1e-12."""
    from cm3d_amd import nusc_io
    g = _g10()
    case = g["cases"][ci]
    res = _g10_dataset(g, case, str(tmp_path))
    tables = nusc_io.NuscTables(g["version"], str(tmp_path))
    cfg = ev.DetectionConfig(case["class_range"], "center_distance", [0.5, 1.0, 2.0, 4.0], 2.0, 0.1, 0.1, 500, 5)
    pred, _ = ev.load_prediction(res, cfg.max_boxes_per_sample)
    gt = ev.load_gt(tables, case["scenes"], rare=case["rare"])
    # the devkit walks nusc.sample in table order; NuscTables keeps that order too
    _same_boxes(ev.add_center_dist(tables, gt), _with_ego(case["gt_loaded"], tables), "load_gt")
    pred = ev.add_center_dist(tables, pred)
    pred = ev.filter_eval_boxes(tables, pred, cfg.class_range, drivable_filtering=case["drivable_filtering"])
    gt = ev.filter_eval_boxes(tables, gt, cfg.class_range, drivable_filtering=case["drivable_filtering"])
    _same_boxes(pred, case["pred_filtered"], "filtered predictions")
    _same_boxes(gt, case["gt_filtered"], "filtered ground truth")
    assert sum(len(v) for v in case["pred_filtered"].values()) < sum(len(v) for v in case["predictions"].values()) * 0.6     # the filters bite
    for th, want in case["object"].items():
        md, rec = ev.accumulate_object_class(gt, pred, None, float(th))
        assert abs(rec - want["recall_actual"]) < 1e-15
        _same_md(md, want["md"], f"object @ {th}")
    n_empty = 0
    for key, want in case["per_class"].items():
        name, th = key.split(":")
        rec, md = ev.accumulate_with_recall(gt, pred, name, None, float(th))
        if want["recall_actual"] is None:          # the reference returns a bare no_predictions() there (:744, :823); ours adds recall 0
            assert rec == 0
            n_empty += 1
        else:
            assert abs(rec - want["recall_actual"]) < 1e-15, key
        _same_md(md, want["md"], key)
    assert n_empty < len(case["per_class"]) // 2


def _with_ego(gt_loaded, tables):
    """The generator serialised load_gt's boxes before add_center_dist: fill in the ego translation the next reference step
    (add_center_dist, :103-127) computes -- box translation minus the LIDAR_TOP ego pose of the sample."""
    out = {}
    for tok, rows in gt_loaded.items():
        pose = tables.get('ego_pose', tables.get('sample_data', tables.sample_data_of[tok]['LIDAR_TOP'])['ego_pose_token'])['translation']
        out[tok] = [dict(b, ego_translation=[b["translation"][i] - pose[i] for i in range(3)]) for b in rows]
    return out
