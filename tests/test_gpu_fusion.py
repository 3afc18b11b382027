"""GPU: cm3d_bev_match (reference src/nuscenes/linear_matching.py:53-121,231-259) through the C-ABI
against the oracle -- match indices bit-exact, IoUs equal -- and the fusion entry point end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_fusion_host import _rand_boxes, _obj

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(oracle, preds, gts, thr=0.2):
    from cm3d_amd import ops
    got = ops.bev_match(preds, gts, thr)
    assert len(got) == len(preds)
    n_match = 0
    for f, (p, g) in enumerate(zip(preds, gts)):
        pm, gm, iou, _ = oracle.bev_match(ops.match_records(p), ops.match_records(g), thr)
        ids = np.flatnonzero(pm >= 0)
        assert np.array_equal(got[f][0], ids), f"sample {f}: matched predictions"
        assert np.array_equal(got[f][1], pm[ids]), f"sample {f}: matched sam3d boxes"
        assert np.array_equal(got[f][2], iou[ids]), f"sample {f}: IoUs"
        n_match += ids.size
    return n_match


def test_bev_match_equals_oracle_on_random_samples(oracle):
    rng = np.random.default_rng(21)
    shapes = [(1, 1), (1, 9), (9, 1), (5, 5), (12, 30), (30, 12), (64, 64), (65, 63), (130, 200), (0, 4), (4, 0), (0, 0), (200, 90)]
    preds, gts = [], []
    for P, G in shapes:
        centre = rng.uniform(-1, 1, 2) * (600.0, 1600.0)              # global-frame magnitudes
        spread = 2.0 + 0.6 * np.sqrt(max(P, G))
        preds.append(_rand_boxes(rng, P, centre, spread))
        gts.append(_rand_boxes(rng, G, centre, spread))
    gts[3][2] = 0.0                                                    # zeros(D) = "no box" (:65)
    preds[4][5] = gts[4][7]                                            # an identical pair: IoU 1
    assert _check(oracle, preds, gts) > 150
    assert _check(oracle, preds, gts, thr=0.05) > 200
    assert _check(oracle, preds, gts, thr=0.7) >= 1


def test_bev_match_noisy_copies_and_many_samples(oracle):
    """SAM3D-like input: the gt side is a noisy, shuffled superset of the predictions; 1500 samples in one call."""
    rng = np.random.default_rng(22)
    preds, gts = [], []
    for f in range(1500):
        P = int(rng.integers(1, 40))
        p = _rand_boxes(rng, P, rng.uniform(-1, 1, 2) * (700.0, 1500.0), 25.0)
        g = p.copy()
        g[:, :2] += rng.normal(0, 0.5, (P, 2)); g[:, 3:5] *= rng.uniform(0.8, 1.2, (P, 2)); g[:, 6] += rng.normal(0, 0.2, P)
        g = np.concatenate([g[rng.random(P) < 0.8], _rand_boxes(rng, int(rng.integers(0, 30)), p[0, :2], 25.0)])
        rng.shuffle(g)
        preds.append(p); gts.append(g)
    n = _check(oracle, preds, gts)
    assert n > 8000


def test_bev_match_large_sample(oracle):
    rng = np.random.default_rng(23)
    preds, gts = [_rand_boxes(rng, 1024, (0, 0), 60.0)], [_rand_boxes(rng, 700, (0, 0), 60.0)]
    assert _check(oracle, preds, gts) > 100


def test_bev_match_capacity_is_reported():
    from cm3d_amd import ops, _lib
    rng = np.random.default_rng(24)
    with pytest.raises(_lib.Cm3dError):
        ops.bev_match([_rand_boxes(rng, 1025)], [_rand_boxes(rng, 3)])


def test_fusion_entry_point(tmp_path, oracle):
    """src/nuscenes/linear_matching.py on the synthetic dataset: predictions = the lifted pseudo-labels, SAM3D file =
    perturbed ground truth with low raw scores.  The best alpha's file equals fuse() with the oracle's matches."""
    from cm3d_amd import fusion, nusc_io, ops, synthetic as syn
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=2, frames_per_scene=3)
    out_dir = tmp_path / "outputs"
    env = dict(os.environ, CM3D_VER_NAME="v1.0-synth", CM3D_INPUT_PATH=dataroot, CM3D_INPUT_DIR=mask_dir, CM3D_OUTPUT_DIR=str(out_dir))
    r = subprocess.run([sys.executable, "2d_to_3d.py", "--ratio", str(cfg.ratio)], cwd=os.path.join(ROOT, "src", "nuscenes"), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    pred_path = out_dir / "pseudolabels_minival.json"
    pred = json.load(open(pred_path))
    tables = nusc_io.NuscTables("v1.0-synth", dataroot)
    rng = np.random.default_rng(7)
    sam = {"meta": pred["meta"], "results": {}}
    for ann in tables.t["sample_annotation"].values():
        tok = ann["sample_token"]
        if rng.random() < 0.25:
            continue
        t = np.asarray(ann["translation"]) + rng.normal(0, 0.15, 3)
        sam["results"].setdefault(tok, []).append(
            {"sample_token": tok, "translation": t.tolist(), "size": list(ann["size"]), "rotation": list(ann["rotation"]), "velocity": [0, 0],
             "detection_name": "car", "detection_score": float(np.round(rng.uniform(0.05, 0.5), 3)), "attribute_name": ""})
    sam_path = tmp_path / "sam3d.json"
    json.dump(sam, open(sam_path, "w"))
    env.update(CM3D_PRED_JSON=str(pred_path), CM3D_SAM3D_JSON=str(sam_path), CM3D_MATCHED_JSON=str(out_dir / "matched.json"),
               CM3D_BEST_JSON=str(out_dir / "best.json"))
    r = subprocess.run([sys.executable, "linear_matching.py"], cwd=os.path.join(ROOT, "src", "nuscenes"), env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Best Score:" in r.stdout and "best alpha" in r.stdout
    best_alpha = float(r.stdout.strip().splitlines()[-1].split("best alpha")[1].split(",")[0])
    best = json.load(open(out_dir / "best.json"))
    # expectation: same host logic, matches from the oracle
    pb, ps, _, _ = fusion.parse_results(pred["results"])
    sb, ss, _, _ = fusion.parse_results(sam["results"], zero_min_quirk=True)
    pm, sm = {}, {}
    n_matched = 0
    for ts in pb:
        pm[ts], sm[ts] = [], []
        if ts in sb and len(pb[ts]) and len(sb[ts]):
            a, _, _, _ = oracle.bev_match(ops.match_records(np.array(pb[ts])), ops.match_records(np.array(sb[ts])), 0.2)
            ids = np.flatnonzero(a >= 0)
            pm[ts], sm[ts] = [int(i) for i in ids], [int(a[i]) for i in ids]
            n_matched += len(ids)
    want, counts = fusion.fuse(pb, ps, sb, ss, pm, sm, best_alpha)
    assert best == json.loads(json.dumps(want))
    assert counts["num_matched_boxes"] == n_matched and f"num_matched_boxes {n_matched}" in r.stdout


def test_integration_md_fusion_binding_runs_as_written(oracle):
    """The `match` stub printed in INTEGRATION.md for linear_matching.py, executed verbatim (library path filled in)."""
    import re
    from cm3d_amd import _lib, ops
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if "cm3d_bev_match" in b][0].replace('ctypes.CDLL("libcm3d_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(31)
    pred, gt = _rand_boxes(rng, 23, (640.0, 1600.0), 6.0), _rand_boxes(rng, 40, (640.0, 1600.0), 6.0)
    ids, gids, ious = ns["match"](pred, gt, 0.2)
    pm, gm, iou, _ = oracle.bev_match(ops.match_records(pred), ops.match_records(gt), 0.2)
    want = np.flatnonzero(pm >= 0)
    assert want.size > 3 and np.array_equal(ids, want) and np.array_equal(gids, pm[want]) and np.array_equal(ious, iou[want])


def test_bev_match_ties_and_kernel_boundaries(oracle):
    """Tie-heavy samples (duplicated boxes: many equal weights, many equally good assignments) at the sizes where the
    solver changes kernels (64 / 128 boxes per side): the three device kernels and the oracle must take the same steps."""
    rng = np.random.default_rng(33)
    preds, gts = [], []
    for P, G in [(63, 64), (64, 64), (65, 64), (64, 65), (127, 128), (128, 128), (129, 128), (128, 129), (40, 200), (200, 40), (130, 131)]:
        base = _rand_boxes(rng, 12, (100.0, -50.0), 8.0)
        p = base[rng.integers(0, 12, P)].copy()
        g = base[rng.integers(0, 12, G)].copy()
        g[rng.random(G) < 0.3, :2] += 0.25                     # some shifted copies: a second weight level
        preds.append(p); gts.append(g)
    assert _check(oracle, preds, gts) > 400
    assert _check(oracle, preds, gts, thr=0.9) > 100
