"""GPU: the drop-in entry point (src/nuscenes/2d_to_3d.py) on a synthetic dataset laid out in the
reference's on-disk formats, against the oracle run over the same files."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nuscenes_entry_point_end_to_end(tmp_path, oracle):
    from cm3d_amd import nusc_io, synthetic as syn
    from tests.helpers import oracle_results
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=2, frames_per_scene=3)
    # a frame without mask files: the reference would raise; with --missing-ok it yields no boxes
    os.remove(os.path.join(mask_dir, names[1], "2_masks.pkl"))
    out_dir = tmp_path / "outputs"
    env = dict(os.environ, CM3D_VER_NAME="v1.0-synth", CM3D_INPUT_PATH=dataroot, CM3D_INPUT_DIR=mask_dir, CM3D_OUTPUT_DIR=str(out_dir))
    for script in ("2d_to_3d.py", "2d_to_3d_new.py"):
        r = subprocess.run([sys.executable, script, "--ratio", str(cfg.ratio), "--missing-ok"], cwd=os.path.join(ROOT, "src", "nuscenes"),
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "wrote 6 samples." in r.stdout
    got = json.load(open(out_dir / "pseudolabels_minival.json"))
    assert got["meta"] == {"use_camera": True, "use_lidar": False, "use_radar": False, "use_map": True, "use_external": False}

    tables = nusc_io.NuscTables("v1.0-synth", dataroot)
    exp = {}
    for name in names:
        scene = tables.scene_by_name(name)
        frames = nusc_io.frames_of_scene(tables, scene, mask_dir, n_sweeps=3, ratio=cfg.ratio, missing_ok=True)
        lanes = [nusc_io.load_lane_points(dataroot, tables.location(scene))]
        exp.update(oracle_results(oracle, frames, lanes, [0] * len(frames)))
    assert list(got["results"]) == list(exp)
    n_boxes = 0
    for tok, boxes in exp.items():
        g = got["results"][tok]
        assert len(g) == len(boxes)
        for a, b in zip(g, boxes):
            assert a["detection_name"] == b["detection_name"] and a["detection_score"] == b["detection_score"]
            assert a["size"] == b["size"] and a["attribute_name"] == b["attribute_name"] and a["velocity"] == [0, 0]
            assert np.allclose(a["translation"], b["translation"], rtol=0, atol=1e-4)
            assert np.allclose(a["rotation"], b["rotation"], rtol=0, atol=1e-4)
            n_boxes += 1
    assert n_boxes > 10
    last = tables.samples_of_scene(tables.scene_by_name(names[1]))[2]["token"]
    assert got["results"][last] == []
