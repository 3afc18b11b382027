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
    # the second run prepares its batches in two reader processes (one scene per batch: CM3D_SCENES_PER_BATCH)
    for script, extra in (("2d_to_3d.py", []), ("2d_to_3d_new.py", ["--workers", "2", "--scenes-per-batch", "1"])):
        r = subprocess.run([sys.executable, script, "--ratio", str(cfg.ratio), "--missing-ok"] + extra, cwd=os.path.join(ROOT, "src", "nuscenes"),
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
    # f3: the evaluation entry point on the file the lifting entry point just wrote
    r = subprocess.run([sys.executable, "eval_custom.py", str(out_dir / "pseudolabels_minival.json"), "--output_dir", str(tmp_path / "metrics"),
                        "--dataroot", dataroot, "--version", "v1.0-synth", "--object_only", "1"],
                       cwd=os.path.join(ROOT, "src", "nuscenes"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "NDS:" in r.stdout and "object" in r.stdout
    summary = json.load(open(tmp_path / "metrics" / "metrics_summary.json"))
    assert 0.0 < summary["mean_ap"] <= 1.0


def _write_waymo_scene(tmp_path, scene, first, count):
    """Extracted-frame files + mask files of one synthetic Waymo scene (frames first .. first+count-1 of the generator)."""
    import pickle
    from cm3d_amd import geometry as geo, synthetic as syn
    cfg = syn.config("tiny", n_cams=5)
    fdir, mdir = tmp_path / "frames" / scene, tmp_path / "masks" / scene
    os.makedirs(fdir); os.makedirs(mdir)
    centre = None
    for i in range(count):
        fr = syn.make_waymo_frame(cfg, first + i)
        P = np.asarray(fr.pose).reshape(4, 4)
        centre = P[:2, 3] if centre is None else centre
        # recover the raw calibration the frame was built from (the generator keeps it in the record only)
        S = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)
        base = syn.make_frame(cfg, first + i)
        ext, intr = [], []
        for c in range(base.cams.shape[0]):
            t_cs_neg, R_csT, _ = geo.cam_stage(base.cams[c], 1)
            T = np.eye(4); T[:3, :3] = R_csT.T; T[:3, 3] = -t_cs_neg
            K = geo.cam_K(base.cams[c]) / cfg.ratio
            ext.append((T @ S).reshape(16)); intr.append([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, 0, 0, 0])
        rec = dict(points=fr.sweeps_raw[0][:, :3], extrinsics=np.array(ext), intrinsics=np.array(intr), pose=P.reshape(16),
                   timestamp_micros=np.int64(fr.timestamp_micros), context_name=np.str_(fr.context_name))
        if i == 0:
            polys = [np.cumsum(np.concatenate([[[centre[0] - 200 + 40 * k, centre[1] - 200, 0.0]], np.tile([[0.0, 0.5, 0.0]], (800, 1))]), 0) for k in range(10)]
            rec["lanes"] = np.vstack(polys); rec["lane_off"] = np.concatenate([[0], np.cumsum([len(p) for p in polys])])
        np.savez_compressed(fdir / f"{i}_frame.npz", **rec)
        pickle.dump(fr.rles, open(mdir / f"{i}_masks.pkl", "wb"))
        json.dump({"labels": fr.labels, "detection_scores": fr.scores, "cam_nums": fr.cam_nums}, open(mdir / f"{i}_data.json", "w"))


def test_waymo_entry_point(tmp_path, oracle):
    """src/waymo/2d_to_3d.py on extracted-frame files, against the oracle on the same files."""
    from cm3d_amd import lifting, pipeline_waymo as pw, waymo as wm
    from tests.helpers import oracle_batch
    scene = "segment-synthetic-0"
    _write_waymo_scene(tmp_path, scene, 0, 3)
    out = tmp_path / "out" / "pred.bin"
    r = subprocess.run([sys.executable, "2d_to_3d.py", "--frames-dir", str(tmp_path / "frames"), "--mask-dir", str(tmp_path / "masks"),
                        "--output", str(out)], cwd=os.path.join(ROOT, "src", "waymo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    blob = open(out, "rb").read()
    # expectation: the oracle over the frames as the entry point loads them (ratio = 1024/1920 like the reference)
    frames, lanes = pw.load_scene(str(tmp_path / "frames"), str(tmp_path / "masks"), scene)
    classes = lifting.ClassTable.waymo()
    hb = lifting.pack_frames(frames, [lanes], [0] * len(frames), classes)
    exp = oracle_batch(oracle, frames, [lanes], [0] * len(frames), hb)
    exp_objs = wm.objects_from_results(hb, exp, classes, [(f.context_name, f.timestamp_micros) for f in frames])
    assert f"wrote {len(exp_objs)} objects" in r.stdout
    got, want = wm.decode_objects(blob), wm.decode_objects(wm.encode_objects(exp_objs))
    assert len(got) == len(want) > 0
    for a, b in zip(got, want):
        assert (a["type"], a["id"], a["context_name"], a["timestamp_micros"], a["score"]) == (b["type"], b["id"], b["context_name"], b["timestamp_micros"], b["score"])
        assert (a["width"], a["length"], a["height"]) == (b["width"], b["length"], b["height"])
        assert np.allclose(a["center"] + [a["heading"]], b["center"] + [b["heading"]], rtol=0, atol=1e-4)


def test_kitti_entry_point(tmp_path, oracle):
    """src/kitti/2d_to_3d.py on KITTI-layout files (velodyne .bin, calib .txt, mask pkl/json)."""
    import pickle
    from cm3d_amd import synthetic as syn
    cfg = syn.config("tiny", width=320, height=96, ratio=0.2, n_masks=10)
    kdir, mdir = tmp_path / "kitti", tmp_path / "masks"
    for d in (kdir / "training" / "velodyne", kdir / "training" / "calib", mdir):
        os.makedirs(d)
    for i in range(3):
        fr, cal = syn.make_kitti_frame(cfg, i)
        fr.sweeps_raw[0].astype(np.float32).tofile(kdir / "training" / "velodyne" / f"{i:06d}.bin")
        with open(kdir / "training" / "calib" / f"{i:06d}.txt", "w") as fh:
            for k, v in cal.items():
                fh.write(f"{k}: " + " ".join(repr(float(x)) for x in np.asarray(v).reshape(-1)) + "\n")
        pickle.dump(fr.rles, open(mdir / f"{i}_masks.pkl", "wb"))
        json.dump({"labels": fr.labels, "detection_scores": fr.scores}, open(mdir / f"{i}_data.json", "w"))
    r = subprocess.run([sys.executable, "2d_to_3d.py", "--kitti-dir", str(kdir), "--mask-dir", str(mdir), "--ratio", str(cfg.ratio)],
                       cwd=os.path.join(ROOT, "src", "kitti"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # expectation: the oracle over the frames as the entry point loads them, through the same label writer
    from cm3d_amd import kitti as kt, lifting
    from tests.helpers import oracle_batch
    frames = []
    for i in range(3):
        rles = pickle.load(open(mdir / f"{i}_masks.pkl", "rb"))
        data = json.load(open(mdir / f"{i}_data.json"))
        frames.append(kt.frame_from_files(i, str(kdir / "training" / "velodyne" / f"{i:06d}.bin"), str(kdir / "training" / "calib" / f"{i:06d}.txt"),
                                          rles, data["labels"], data["detection_scores"], cfg.ratio))
    hb = lifting.pack_frames(frames, [[[0.0, 0.0, 0.0]]], [0] * 3)
    exp = oracle_batch(oracle, frames, [np.zeros((1, 3))], [0] * 3, hb)
    per_mask_frame = np.repeat(np.arange(3), np.diff(hb.mask_off))
    exp["hit_xyz"] = exp["points"][np.repeat(exp["pt_off"][per_mask_frame], np.diff(exp["hit_off"])) + exp["hit_idx"]]
    total = 0
    for i in range(3):
        pred = open(kdir / "training" / "pred" / f"{i:06d}.txt").read().splitlines()
        pseudo = open(kdir / "training" / "pseudo" / f"{i:06d}.txt").read().splitlines()
        want_pred, want_pseudo = kt.labels_of_frame(hb, exp, i, lifting.ClassTable.nuscenes(), lifting.SHAPE_PRIORS_CHATGPT)
        assert pred == [l.rstrip("\n") for l in want_pred] and pseudo == [l.rstrip("\n") for l in want_pseudo], i
        for a, b in zip(pred, pseudo):
            assert a.rsplit(" ", 1)[0] == b and len(a.split()) == 16 and a.split()[0] in kt.KITTI_CLASS_MAPS.values()
        total += len(pred)
    assert total > 3 and f"wrote {total} labels" in r.stdout


def test_g7_tiny_scene_json(tmp_path):
    """G7: the committed end-to-end fixture (a 1-scene, 2-frame synthetic dataset through the oracle pipeline)
    against the entry point's output on the regenerated dataset."""
    from cm3d_amd import nusc_io, synthetic as syn
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=1, frames_per_scene=2)
    out_dir = tmp_path / "outputs"
    env = dict(os.environ, CM3D_VER_NAME="v1.0-synth", CM3D_INPUT_PATH=dataroot, CM3D_INPUT_DIR=mask_dir, CM3D_OUTPUT_DIR=str(out_dir))
    r = subprocess.run([sys.executable, "2d_to_3d.py", "--ratio", str(cfg.ratio)], cwd=os.path.join(ROOT, "src", "nuscenes"), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.load(open(out_dir / "pseudolabels_minival.json"))
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "g7_tiny_scene.json")))
    assert got["meta"] == want["meta"] and list(got["results"]) == list(want["results"])
    n = 0
    for tok in want["results"]:
        assert len(got["results"][tok]) == len(want["results"][tok])
        for a, b in zip(got["results"][tok], want["results"][tok]):
            assert {k: a[k] for k in a if k not in ("translation", "rotation")} == {k: b[k] for k in b if k not in ("translation", "rotation")}
            assert np.allclose(a["translation"], b["translation"], rtol=0, atol=1e-4) and np.allclose(a["rotation"], b["rotation"], rtol=0, atol=1e-4)
            n += 1
    assert n > 5


def test_nuscenes_entry_point_two_ranks(tmp_path):
    """The N>1 path of the entry point: two ranks (gloo, both on the one GPU of the box) shard the scenes, rank 0 gathers
    and writes; the file must equal the single-process one."""
    from cm3d_amd import nusc_io, synthetic as syn
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=3, frames_per_scene=2)
    base = dict(os.environ, CM3D_VER_NAME="v1.0-synth", CM3D_INPUT_PATH=dataroot, CM3D_INPUT_DIR=mask_dir)
    cwd = os.path.join(ROOT, "src", "nuscenes")
    r = subprocess.run([sys.executable, "2d_to_3d.py", "--ratio", str(cfg.ratio)], cwd=cwd, env=dict(base, CM3D_OUTPUT_DIR=str(tmp_path / "one")),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(base, CM3D_OUTPUT_DIR=str(tmp_path / "two"), CM3D_DIST_BACKEND="gloo", CM3D_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29721", "2d_to_3d.py", "--ratio", str(cfg.ratio), "--scenes-per-batch", "1"], cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    one = json.load(open(tmp_path / "one" / "pseudolabels_minival.json"))
    two = json.load(open(tmp_path / "two" / "pseudolabels_minival.json"))
    assert list(one["results"]) == list(two["results"]) and len(one["results"]) == 6
    assert one == two and sum(len(v) for v in one["results"].values()) > 10


def test_waymo_entry_point_two_ranks(tmp_path):
    """The N>1 path of the Waymo entry point: two ranks (gloo, both on the one GPU of the box) shard the scenes, one
    exchange of box records, rank 0 writes; the file must equal the single-process one byte for byte."""
    for k in range(3):
        _write_waymo_scene(tmp_path, f"segment-synthetic-{k}", 10 * k, 2)
    cwd = os.path.join(ROOT, "src", "waymo")
    args = ["--frames-dir", str(tmp_path / "frames"), "--mask-dir", str(tmp_path / "masks")]
    r = subprocess.run([sys.executable, "2d_to_3d.py", *args, "--output", str(tmp_path / "one.bin")], cwd=cwd, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, CM3D_DIST_BACKEND="gloo", CM3D_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29722", "2d_to_3d.py", *args, "--output", str(tmp_path / "two.bin")], cwd=cwd, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    one, two = open(tmp_path / "one.bin", "rb").read(), open(tmp_path / "two.bin", "rb").read()
    from cm3d_amd import waymo as wm
    assert one == two and len(wm.decode_objects(one)) > 5
