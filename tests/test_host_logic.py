"""CPU: host-side logic of the product path (no kernels run here)."""
import json
import os
import re

import numpy as np
import pytest

from cm3d_amd import geometry as geo, rle, synthetic as syn


def test_rle_codec_matches_oracle_codec(oracle):
    rng = np.random.default_rng(0)
    for _ in range(30):
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        img = (rng.uniform(size=(H, W)) < rng.uniform(0.05, 0.9)).astype(np.uint8)
        if rng.uniform() < 0.2:
            img[:] = rng.integers(0, 2)
        cnts = rle.dense_to_counts(img)
        assert int(cnts.sum()) == H * W
        s = rle.counts_to_string(cnts)
        ref = oracle.rle_encode(img.T)           # the producer encodes the (W,H) transpose, F-ordered
        assert ref["size"] == [W, H] and ref["counts"] == s
        assert np.array_equal(rle.string_to_counts(s), cnts)
        assert np.array_equal(rle.counts_to_dense(cnts, W, H), img)
        assert np.array_equal(oracle.rle_decode({"size": [W, H], "counts": s}).T, img)


def test_rle_large_runs_and_negative_deltas():
    cnts = np.array([0, 5, 1000000, 3, 7, 2000000, 1, 1], np.uint32)
    s = rle.counts_to_string(cnts)
    assert np.array_equal(rle.string_to_counts(s), cnts)
    assert rle.string_to_counts(b"").size == 0
    with pytest.raises(ValueError):
        rle.string_to_counts(b"P")          # continuation bit (0x20) set on the last character


def test_spans_to_counts_equals_dense():
    rng = np.random.default_rng(1)
    W, H = 64, 20
    rows = np.arange(3, 15)
    x0 = rng.integers(0, 30, rows.size)
    x1 = x0 + rng.integers(0, 34, rows.size)
    x0[2], x1[2] = 0, W - 1            # a full row
    x1[5] = W - 1
    x0[6] = 0                          # touches the previous span in linear order
    img = np.zeros((H, W), np.uint8)
    for y, a, b in zip(rows, x0, x1):
        img[y, a:b + 1] = 1
    assert np.array_equal(rle.spans_to_counts(rows, x0, x1, W, H), rle.dense_to_counts(img))
    assert np.array_equal(rle.spans_to_counts([], [], [], W, H), [W * H])


def test_quaternion_helpers():
    rng = np.random.default_rng(2)
    for _ in range(50):
        q = rng.normal(size=4)
        R = geo.quat_to_rotmat(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1.0)
        q2 = geo.rotmat_to_quat(R)
        assert np.allclose(geo.quat_to_rotmat(q2), R, atol=1e-12)
    assert np.allclose(geo.quat_to_rotmat([1, 0, 0, 0]), np.eye(3))
    assert np.allclose(geo.quat_to_rotmat([np.cos(0.3), 0, 0, np.sin(0.3)]), geo.rot_z(0.6))


def test_scaled_intrinsic_is_float32_product():
    K = np.array([[1266.417203046554, 0.0, 816.2670197447984], [0.0, 1266.417203046554, 491.50706579294757], [0, 0, 1.0]])
    Ks = geo.scaled_intrinsic_f32(K, 0.64)
    assert Ks.dtype == np.float32 and Ks[2, 2] == 1.0
    assert Ks[0, 0] == np.float32(np.float32(K[0, 0]) * np.float32(0.64))


def test_synthetic_is_deterministic_and_well_formed():
    cfg = syn.config("tiny")
    a, b = syn.make_frame(cfg, 7), syn.make_frame(cfg, 7)
    assert all(np.array_equal(x, y) for x, y in zip(a.sweeps_raw, b.sweeps_raw))
    assert [r["counts"] for r in a.rles] == [r["counts"] for r in b.rles]
    assert len(a.rles) == len(a.labels) == len(a.scores) == len(a.cam_nums) == cfg.n_masks
    assert all(r["size"] == [cfg.width, cfg.height] for r in a.rles)
    assert all(0 <= c < cfg.n_cams for c in a.cam_nums)
    assert a.sweeps_raw[0].shape == (cfg.n_points, 5) and a.sweeps_raw[0].dtype == np.float32


def test_pack_frames_layout_and_validation():
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(3)]
    lanes = [syn.make_lane_table([600, 1600], 500, seed=0)]
    hb = lifting.pack_frames(frames, lanes, [0, 0, 0])
    assert hb.n_frames == 3 and hb.n_masks == 3 * cfg.n_masks
    assert hb.n_real_rows == sum(r.shape[0] for f in frames for r in f.sweeps_raw)
    assert np.array_equal(hb.mask_frame, np.repeat(np.arange(3), cfg.n_masks))
    assert hb.lane.dtype == np.float32 and hb.lane_off.tolist() == [0, 500]
    # labels are renamed like get_detection_name and mapped to the class table
    names = lifting.ClassTable.nuscenes().names
    assert [names[c] for c in hb.class_id[:cfg.n_masks]] == [lifting.get_detection_name(l) for l in frames[0].labels]
    bad = syn.make_frame(cfg, 0)
    bad.labels[0] = "unicorn"
    with pytest.raises(ValueError):
        lifting.pack_frames([bad], lanes, [0])
    bad2 = syn.make_frame(cfg, 0)
    bad2.rles[0] = {"size": [cfg.width, cfg.height], "counts": rle.counts_to_string(np.array([5], np.uint32))}
    with pytest.raises(ValueError):
        lifting.pack_frames([bad2], lanes, [0])


def test_reference_tables():
    from cm3d_amd import lifting
    assert lifting.get_detection_name("trafficcone") == "traffic_cone"
    assert lifting.get_detection_name("constructionvehicle") == "construction_vehicle"
    assert lifting.get_detection_name("human") == "pedestrian"
    assert lifting.get_detection_name("car") == "car"
    ct = lifting.ClassTable.nuscenes()
    assert ct.prior_wlh[ct.index("bus")].tolist() == [2.5, 12.0, 4.0]
    assert ct.nms_thr[ct.index("pedestrian")] == 0.175 and ct.is_vehicle[ct.index("barrier")] == 1
    assert ct.is_vehicle[ct.index("bicycle")] == 0
    assert set(lifting.ATTRIBUTE_NAMES) == set(ct.names)


def test_box_records_schema():
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(2)]
    hb = lifting.pack_frames(frames, [syn.make_lane_table([600, 1600], 100, seed=0)], [0, 0])
    M = hb.n_masks
    res = {"flags": np.zeros(M, np.int32), "box": np.zeros((M, 10))}
    res["flags"][[1, 3]] = 3
    res["flags"][2] = 1              # has a box but was suppressed
    res["box"][1] = [1, 2, 3, 0.5, 0.5, 0, 0, 0, 0, 3]
    out = lifting.box_records(hb, res)
    assert list(out) == hb.tokens and out[hb.tokens[1]] == []
    b = out[hb.tokens[0]]
    assert len(b) == 2 and b[0]["translation"] == [1.0, 2.0, 3.0] and b[0]["rotation"] == [0.5, 0.0, 0.0, 0.5]
    assert set(b[0]) == {"sample_token", "translation", "size", "rotation", "velocity", "detection_name",
                         "detection_score", "attribute_name"}
    json.dumps(out)


def test_shard_helpers():
    from cm3d_amd import dist
    for n, w in [(10, 4), (3, 8), (28130, 8), (1, 1)]:
        r = [dist.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    sizes = [40, 41, 39, 40, 38, 40, 41, 40, 39, 40, 40]
    b = dist.shard_scenes(sizes, 4)
    assert b[0][0] == 0 and b[-1][1] == len(sizes) and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    loads = [sum(sizes[a:c]) for a, c in b]
    assert max(loads) <= 1.5 * sum(sizes) / 4


def test_c3_partition_covers_trainval_exactly_once():
    """BASELINE config C3: the 28 130 keyframes / 850 scenes of nuScenes-trainval over 8 ranks (also 1, 2, 4): every frame
    and every scene is owned by exactly one rank, in order, and the frame loads are balanced."""
    from cm3d_amd import dist
    rng = np.random.default_rng(5)
    sizes = rng.integers(26, 41, 850)                         # nuScenes scenes hold ~33 keyframes on average
    while sizes.sum() != 28130:                               # total as in BASELINE.md
        k = rng.integers(0, 850)
        sizes[k] += 1 if sizes.sum() < 28130 else -1
    assert sizes.sum() == 28130 and sizes.min() > 0
    starts = np.concatenate([[0], np.cumsum(sizes)])
    for world in (1, 2, 4, 8):
        owner = np.full(28130, -1)
        for rank in range(world):
            a, b = dist.shard_range(28130, rank, world)
            assert np.all(owner[a:b] == -1)
            owner[a:b] = rank
        assert np.all(owner >= 0) and np.all(np.diff(owner) >= 0)
        counts = np.bincount(owner, minlength=world)
        assert counts.max() - counts.min() <= 1
        bounds = dist.shard_scenes([int(x) for x in sizes], world)
        assert len(bounds) == world and bounds[0][0] == 0 and bounds[-1][1] == 850
        assert all(x[1] == y[0] for x, y in zip(bounds, bounds[1:])) and all(lo < hi for lo, hi in bounds)
        loads = np.array([starts[hi] - starts[lo] for lo, hi in bounds])
        assert loads.sum() == 28130 and loads.max() <= 28130 / world + 42          # within one scene of the ideal share
        # more ranks than scenes: trailing ranks own nothing, nothing is lost
        few = dist.shard_scenes([40, 40, 40], 8)
        assert [hi - lo for lo, hi in few].count(1) == 3 and sum(hi - lo for lo, hi in few) == 3


def test_gathered_records_rebuild_the_box_dicts():
    """The single exchange ships kept-box records (lifting.kept_box_records); rank 0 rebuilds the output dicts from them and
    the job's token order (lifting.nuscenes_boxes_from_records): same result as building them from the per-mask arrays."""
    import torch
    from types import SimpleNamespace
    from cm3d_amd import lifting
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(3)]
    hb = lifting.pack_frames(frames, [syn.make_lane_table([600, 1600], 100, seed=0)], [0, 0, 0])
    M = hb.n_masks
    rng = np.random.default_rng(1)
    flags = rng.choice([0, 1, 3], M).astype(np.int32)
    box = rng.normal(0, 50, (M, 10))
    box[:, 7], box[:, 8], box[:, 9] = hb.score, hb.class_id, flags
    want = lifting.box_records(hb, {"flags": flags, "box": box})
    # the job has two more samples in front of this batch's three (records carry the global sample index)
    tokens = ["other-0", "other-1"] + hb.tokens
    b = SimpleNamespace(flags=torch.from_numpy(flags), box=torch.from_numpy(box), mask_frame=torch.from_numpy(hb.mask_frame))
    rec = lifting.kept_box_records(b, np.array([[2 + f, 0] for f in range(3)], np.float64))
    assert rec.shape == (int((flags == 3).sum()), 10) and rec.dtype == torch.float64
    # split over two "ranks" and concatenated in rank order, as gather_records returns them
    parts = [rec[:4], rec[4:]]
    got = lifting.nuscenes_boxes_from_records(torch.cat(parts, 0).numpy(), tokens)
    assert list(got) == tokens and got["other-0"] == [] and got["other-1"] == []
    assert {t: got[t] for t in hb.tokens} == want
    json.dumps(got)


def test_direct_json_writer_equals_json_dumps_of_the_dicts():
    """lifting.nuscenes_results_json writes the result file's text straight from the box records; it must be, character for
    character, json.dumps of the reference-shaped dicts (key order, separators, float repr, Infinity / NaN spelling, escaped
    tokens, samples without boxes)."""
    from cm3d_amd import lifting
    rng = np.random.default_rng(12)
    for n, nt in ((0, 3), (1, 1), (40, 5), (3000, 64)):
        rec = rng.normal(size=(n, 10)) * rng.choice([1e-9, 1.0, 1e3, 1e17], size=(n, 1))
        if n:
            rec[:, lifting.REC_FRAME_A] = rng.integers(0, nt, n)
            rec[:, 8] = rng.integers(0, 10, n)
            rec[0, 0] = np.inf
            rec[n // 2, 7] = np.nan
            rec[n - 1, 4] = -0.0
        tokens = [f'tok"{i}\\' for i in range(nt)]
        meta = {"use_camera": False, "use_lidar": True}
        want = json.dumps({"meta": meta, "results": lifting.nuscenes_boxes_from_records(rec, tokens)})
        got, n_boxes = lifting.nuscenes_results_json(rec, tokens, meta=meta)
        assert got == want and n_boxes == n
        assert json.loads(got.replace("Infinity", "1e999").replace("NaN", "null"))["results"].keys() == set(tokens)


def test_prepare_scene_batch_in_reader_processes(tmp_path):
    """pipeline_nuscenes.prepare_scene_batch (file reads + RLE strings + packing; no GPU) gives the same host batches in
    spawned reader processes as in this process."""
    import multiprocessing as mp
    import numpy as np
    from cm3d_amd import nusc_io, pipeline_nuscenes as pn, synthetic as syn
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=2, frames_per_scene=2)
    tasks = [("v1.0-synth", dataroot, mask_dir, [n], 3, cfg.ratio, False, None) for n in names]
    here = [pn.prepare_scene_batch(t) for t in tasks]
    with mp.get_context("spawn").Pool(2) as pool:
        there = list(pool.imap(pn.prepare_scene_batch, [t + (True,) for t in tasks]))       # sweeps through shared memory
    keep = []
    for (tok_a, hbs_a, _), (tok_b, hbs_b, _) in zip(here, there):
        assert tok_a == tok_b and len(tok_a) == 2 and len(hbs_a) == len(hbs_b) == 1
        a, b = hbs_a[0], hbs_b[0]
        assert isinstance(b.raw, tuple)
        pn._attach_raw(b, keep)
        assert a.tokens == b.tokens and a.labels == b.labels and (a.width, a.height, a.n_cams) == (b.width, b.height, b.n_cams)
        for k in ("raw", "sweep_row_off", "sweep_xf", "frame_sweep_off", "cams", "mask_off", "mask_cam", "rle_counts", "rle_off", "class_id",
                  "score", "lane", "lane_off", "frame_lane", "ego_xyz"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        b.raw = None
    names_in_shm = [s.name for s in keep]
    pn._release(keep)
    assert len(names_in_shm) == 2 and not any(os.path.exists("/dev/shm/" + n.lstrip("/")) for n in names_in_shm)


def test_two_pass_medoid_error_bound_holds_numerically():
    """The bound k_medoid_long relies on (cm3d_amd/csrc/medoid.hip): float32 sums accumulated in the same
    order, terms perturbed by at most one ulp (all up, all down, or at random -- worse than any real v_sqrt_f32), and terms
    below 1e-15 replaced by zero, differ by at most E = 1.01 (M + 2) 2^-23 A + 2 M 1e-15."""
    rng = np.random.default_rng(12)
    for M in (513, 5000, 60000):
        for scale in (0.05, 3.0, 60.0):
            t = np.sqrt(rng.uniform(0, scale, M).astype(np.float32) ** 2).astype(np.float32)
            t[rng.integers(0, M, M // 50)] = 0.0                                   # the diagonal and duplicates
            tiny = rng.integers(0, M, M // 100)
            t[tiny] = np.float32(5e-16)                                            # roots of d2 < 1e-30
            S = np.cumsum(t, dtype=np.float32)[-1]
            for mode in ("up", "down", "rand"):
                d = {"up": np.ones(M), "down": -np.ones(M), "rand": rng.choice([-1.0, 0.0, 1.0], M)}[mode]
                tp = np.where(d > 0, np.nextafter(t, np.float32(np.inf)), np.where(d < 0, np.nextafter(t, np.float32(-np.inf)), t))
                tp = np.maximum(tp, 0).astype(np.float32)
                tp[tiny] = 0.0
                A = np.cumsum(tp, dtype=np.float32)[-1]
                E = 1.01 * (M + 2) * 2.0 ** -23 * float(A) + 2.0 * M * 1e-15
                assert abs(float(A) - float(S)) <= E, (M, scale, mode, float(A), float(S), E)


def test_kitti_obb_yaw_known_answers():
    """KITTI yaw (src/kitti/2d_to_3d.py:855-876,1524): PCA box of the hull vertices, axes re-ordered by extent, euler 'zyx'[0].
    Known answers up to the eigenvector signs, which Open3D's solver and LAPACK need not share: an
    axis-aligned box of points has yaw 0 (mod pi), the same box turned by 30 degrees about z has yaw +-30 degrees (mod pi);
    flat or collinear points make Qhull fail -> the caller's identity fallback (:1481-1484), yaw 0."""
    from cm3d_amd import kitti as kt
    rng = np.random.default_rng(4)
    box = rng.uniform(-0.5, 0.5, (400, 3)) * [4.2, 1.8, 1.4]                  # long axis x, then y, then z
    box = np.concatenate([box, np.array([[sx * 2.1, sy * 0.9, sz * 0.7] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])])

    def mod_pi(a):
        return (a + np.pi / 2) % np.pi - np.pi / 2

    assert abs(mod_pi(kt.obb_yaw(box + [10.0, -3.0, 25.0]))) < 1e-6
    # turned about z by less than 45 degrees the extents keep their order (x > y > z) and the yaw is the turn -- up to the
    # signs of the eigenvectors: flipping the first one adds pi, flipping the third one negates the angle
    for deg in (30.0, -20.0, 40.0):
        a = np.deg2rad(deg)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        y = kt.obb_yaw(box @ Rz.T + [3.0, 1.0, 12.0])
        assert min(abs(mod_pi(y - a)), abs(mod_pi(y + a))) < 1e-6, (deg, y)
    with pytest.raises(Exception):
        kt.obb_yaw(np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0.0]]))   # collinear: no hull


def test_kitti_label_lines_follow_the_reference():
    """Label file contents (:1523-1536): KITTI class of the label (:105-116,183-197), -1 -1 -10 and a zero 2D box, the prior of
    the label as (h, w, l), the medoid with y moved down by h / 2, yaw, score (pred only)."""
    from types import SimpleNamespace
    from cm3d_amd import kitti as kt, lifting
    rng = np.random.default_rng(2)
    pts = [rng.normal(0, 1, (6, 3)) + [0, 0, 20], rng.normal(0, 1, (3, 3)), rng.normal(0, 1, (5, 3)) + [5, 1, 30]]   # 6, 3 (skipped) and 5 points
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])])
    hb = SimpleNamespace(mask_off=np.array([0, 3]), labels=[["bus", "car", "human"]], score=np.array([0.91, 0.5, 0.33]))
    res = dict(hit_off=off, hit_xyz=np.concatenate(pts).astype(np.float32), centroid=np.array([[1.0, 2.0, 20.0], [0, 0, 0], [5.0, 1.5, 30.0]], np.float32))
    pred, pseudo = kt.labels_of_frame(hb, res, 0, lifting.ClassTable.nuscenes(), lifting.SHAPE_PRIORS_CHATGPT)
    assert len(pred) == len(pseudo) == 2                                       # the 3-point mask writes nothing (:1479-1480)
    f = pred[0].split()
    w, l, h = lifting.SHAPE_PRIORS_CHATGPT["bus"]
    assert f[0] == "Tram" and f[1:8] == ["-1", "-1", "-10", "0", "0", "0", "0"] and [float(v) for v in f[8:11]] == [h, w, l]
    assert [float(v) for v in f[11:14]] == [1.0, 2.0 + h / 2, 20.0] and float(f[15]) == 0.91 and len(f) == 16
    g = pred[1].split()
    assert g[0] == "Pedestrian" and [float(v) for v in g[8:11]] == [1.7, 0.4, 0.7] and pred[1].rsplit(" ", 1)[0] + "\n" == pseudo[1]


def test_waymo_objects_parse_with_the_protobuf_runtime():
    """cm3d_amd.waymo hand-encodes metrics_pb2.Objects (src/waymo/2d_to_3d.py:1034-1065,1300-1305).  waymo-open-dataset is not
    installed, so the messages are declared here from the public protos' field numbers (label.proto: Label.Box 1-7, Label
    box=1 type=3 id=4; metrics.proto: Object object=1 score=2 context_name=4 frame_timestamp_micros=5, Objects objects=1)
    and the real protobuf runtime must parse the bytes back to the values that went in -- wire types, varints, lengths."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from cm3d_amd import waymo as wm
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="cm3d_wod_subset.proto", package="cm3d.wod", syntax="proto2")
    label = fd.message_type.add(name="Label")
    box = label.nested_type.add(name="Box")
    for i, n in enumerate(["center_x", "center_y", "center_z", "width", "length", "height", "heading"], 1):
        box.field.add(name=n, number=i, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    en = label.enum_type.add(name="Type")
    for n, v in [("TYPE_UNKNOWN", 0), ("TYPE_VEHICLE", 1), ("TYPE_PEDESTRIAN", 2), ("TYPE_SIGN", 3), ("TYPE_CYCLIST", 4)]:
        en.value.add(name=n, number=v)
    label.field.add(name="box", number=1, type=F.TYPE_MESSAGE, type_name=".cm3d.wod.Label.Box", label=F.LABEL_OPTIONAL)
    label.field.add(name="type", number=3, type=F.TYPE_ENUM, type_name=".cm3d.wod.Label.Type", label=F.LABEL_OPTIONAL)
    label.field.add(name="id", number=4, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    obj = fd.message_type.add(name="Object")
    obj.field.add(name="object", number=1, type=F.TYPE_MESSAGE, type_name=".cm3d.wod.Label", label=F.LABEL_OPTIONAL)
    obj.field.add(name="score", number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    obj.field.add(name="context_name", number=4, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    obj.field.add(name="frame_timestamp_micros", number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    objs = fd.message_type.add(name="Objects")
    objs.field.add(name="objects", number=1, type=F.TYPE_MESSAGE, type_name=".cm3d.wod.Object", label=F.LABEL_REPEATED)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Objects = message_factory.GetMessageClass(pool.FindMessageTypeByName("cm3d.wod.Objects"))
    assert wm.WAYMO_TYPE == {"unknown": 0, "vehicle": 1, "pedestrian": 2, "sign": 3, "cyclist": 4}          # label.proto Label.Type
    rng = np.random.default_rng(9)
    want = []
    for k in range(40):
        want.append(dict(center=rng.normal(0, 80, 3), length=float(rng.uniform(0.3, 12)), width=float(rng.uniform(0.3, 3)),
                         height=float(rng.uniform(0.5, 4)), heading=float(rng.uniform(-np.pi, np.pi)), type_id=int(rng.integers(0, 5)),
                         score=float(np.float32(rng.uniform(0, 1))), context_name=f"segment-{k}_with_camera_labels",
                         timestamp_micros=int(rng.integers(1 << 40, 1 << 52))))
    blob = wm.encode_objects([wm.encode_object(**w) for w in want])
    msg = Objects()
    assert msg.ParseFromString(blob) == len(blob) and len(msg.objects) == 40
    for o, w in zip(msg.objects, want):
        b = o.object.box
        assert [b.center_x, b.center_y, b.center_z] == list(w["center"]) and (b.length, b.width, b.height, b.heading) == (w["length"], w["width"], w["height"], w["heading"])
        assert o.object.type == w["type_id"] and o.object.id == "unique object tracking ID"          # :1047
        assert o.score == w["score"] and o.context_name == w["context_name"] and o.frame_timestamp_micros == w["timestamp_micros"]
    # and the runtime's own serialisation of the parsed message is the very same bytes (canonical field order)
    assert msg.SerializeToString() == blob
    # our decoder (used by the GPU tests) reads the same values
    for d, w in zip(wm.decode_objects(blob), want):
        assert d["center"] == list(w["center"]) and d["type"] == w["type_id"] and d["timestamp_micros"] == w["timestamp_micros"]


@pytest.mark.parametrize("mag", [0.0, 4000.0, 10000.0])
def test_crafted_boundary_rows_fall_on_both_sides(oracle, mag):
    """tests/magnitude_cases.py crafts rows onto the limits the projection kernel's culling relies on (the image's accept
    limits behind the view wedge, the edges of a mask's bounding box, the minimum depth).  For the GPU test built on them to
    mean anything, the reference arithmetic (the oracle) must put a fair share of every kind on EACH side of its limit."""
    from cm3d_amd import rle
    from tests.magnitude_cases import H, N_EACH, N_KINDS, RECT, W, crafted_frames
    frames, crafted = crafted_frames(mag, n_frames=1)
    fr, rows = frames[0], crafted[0]
    x = fr.sweep_xf[0]
    P = oracle.sweep_prep(rows, x[0:9], x[9:12], x[12:21], x[21:24], np.float32(0.0))
    assert P.shape[0] == rows.shape[0] == fr.cams.shape[0] * N_KINDS * N_EACH
    full = oracle.erode3x3(np.ones((H, W), np.uint8))
    rect = np.zeros((H, W), np.uint8)
    rect[RECT[1]:RECT[3] + 1, RECT[0]:RECT[2] + 1] = 1
    rect = oracle.erode3x3(rect)
    for c in range(fr.cams.shape[0]):
        base = c * N_KINDS * N_EACH
        in_img = np.zeros(P.shape[0], bool); in_img[oracle.points_in_mask(P, fr.cams[c], full)] = True
        in_rect = np.zeros(P.shape[0], bool); in_rect[oracle.points_in_mask(P, fr.cams[c], rect)] = True
        for kind in range(N_KINDS):
            sel = slice(base + kind * N_EACH, base + (kind + 1) * N_EACH)
            inside = (in_rect if kind in (4, 5, 6, 7, 9) else in_img)[sel]
            assert 3 <= int(inside.sum()) <= N_EACH - 3, (mag, c, kind, int(inside.sum()))


def _stub_waymo_devkit(monkeypatch):
    """Stand-ins for `tensorflow.compat.v1` and `waymo_open_dataset` with the handful of members the extraction touches
    (src/waymo/2d_to_3d.py:436-479 of the reference): a 'TFRecord' is a pickled list of per-frame dicts, `Frame.ParseFromString`
    turns one back into an object tree shaped like dataset_pb2.Frame, and the range-image conversion hands back the cloud the
    dict carries.  Third-party behaviour is not under test here -- the plumbing around it is."""
    import pickle
    import sys
    import types
    from types import SimpleNamespace as NS

    class _Item:
        def __init__(self, b): self.b = b
        def numpy(self): return self.b

    tf = types.ModuleType("tensorflow.compat.v1")
    tf.enable_eager_execution = lambda: None
    tf.data = NS(TFRecordDataset=lambda path, compression_type="": [_Item(pickle.dumps(d)) for d in pickle.load(open(path, "rb"))])

    class Frame:
        def ParseFromString(self, b):
            d = pickle.loads(bytes(b))
            self._points = d["points"]
            self.timestamp_micros = d["timestamp_micros"]
            self.pose = NS(transform=list(d["pose"]))
            cal = [NS(name=c + 1, extrinsic=NS(transform=list(d["extrinsics"][c])), intrinsic=list(d["intrinsics"][c])) for c in range(len(d["extrinsics"]))]
            self.context = NS(name=d["context_name"], camera_calibrations=cal[::-1])        # (the proto does not promise name order)
            feats = []
            for poly in d.get("lane_polylines", []):
                feats.append(NS(HasField=lambda k: k == "lane", lane=NS(polyline=[NS(x=p[0], y=p[1], z=p[2]) for p in poly])))
            feats.append(NS(HasField=lambda k: False, lane=None))                            # a map feature that is not a lane
            self.map_features = feats

    fu = types.ModuleType("waymo_open_dataset.utils.frame_utils")
    fu.parse_range_image_and_camera_projection = lambda frame: ("ri", "cp", None, "pose")
    fu.convert_range_image_to_point_cloud = lambda frame, ri, cp, pose, idx, keep: ([frame._points], None)
    mods = {"tensorflow": types.ModuleType("tensorflow"), "tensorflow.compat": types.ModuleType("tensorflow.compat"), "tensorflow.compat.v1": tf,
            "waymo_open_dataset": types.ModuleType("waymo_open_dataset"), "waymo_open_dataset.dataset_pb2": types.ModuleType("waymo_open_dataset.dataset_pb2"),
            "waymo_open_dataset.utils": types.ModuleType("waymo_open_dataset.utils"), "waymo_open_dataset.utils.frame_utils": fu}
    mods["tensorflow"].compat = mods["tensorflow.compat"]; mods["tensorflow.compat"].v1 = tf
    mods["waymo_open_dataset.dataset_pb2"].Frame = Frame
    mods["waymo_open_dataset"].dataset_pb2 = mods["waymo_open_dataset.dataset_pb2"]
    mods["waymo_open_dataset"].utils = mods["waymo_open_dataset.utils"]; mods["waymo_open_dataset.utils"].frame_utils = fu
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)


def test_waymo_tfrecord_route_equals_extracted_frame_route(tmp_path, monkeypatch):
    """The Waymo entry point reads the TFRecords itself where the devkit is installed (like the reference, :436-479), and the
    extracted `<f>_frame.npz` files otherwise; tools/extract_waymo_frames.py writes those files with the same function.  With a
    stand-in devkit: both routes give the same frames, lane table and packed batch, frames without mask files are skipped
    without converting their range images, and the scene is named after the TFRecord file as in the reference."""
    import pickle
    from cm3d_amd import geometry as geo, lifting, pipeline_waymo as pw, waymo as wm
    _stub_waymo_devkit(monkeypatch)
    assert wm.devkit_available()
    cfg = syn.config("tiny", n_cams=5)
    scene = "segment-123_with_camera_labels.tfrecord"
    os.makedirs(tmp_path / "tf"); os.makedirs(tmp_path / "masks" / scene)
    S = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)
    dicts = []
    for i in range(4):
        fr, base = syn.make_waymo_frame(cfg, 20 + i), syn.make_frame(cfg, 20 + i)
        P = np.asarray(fr.pose).reshape(4, 4)
        ext, intr = [], []
        for c in range(base.cams.shape[0]):
            t_cs_neg, R_csT, _ = geo.cam_stage(base.cams[c], 1)
            T = np.eye(4); T[:3, :3] = R_csT.T; T[:3, 3] = -t_cs_neg
            K = geo.cam_K(base.cams[c]) / cfg.ratio
            ext.append((T @ S).reshape(16)); intr.append([K[0, 0], K[1, 1], K[0, 2], K[1, 2], 0, 0, 0, 0, 0])
        d = dict(points=fr.sweeps_raw[0][:, :3].copy(), extrinsics=np.array(ext), intrinsics=np.array(intr), pose=P.reshape(16),
                 timestamp_micros=int(fr.timestamp_micros), context_name=fr.context_name)
        if i == 0:
            d["lane_polylines"] = [np.cumsum(np.concatenate([[[P[0, 3] - 100 + 30 * k, P[1, 3] - 100, 0.0]], np.tile([[0.0, 0.5, 0.0]], (300, 1))]), 0) for k in range(6)]
        dicts.append(d)
        if i != 2:                                   # frame 2 has no detections: the producer wrote no files for it
            pickle.dump(fr.rles, open(tmp_path / "masks" / scene / f"{i}_masks.pkl", "wb"))
            json.dump({"labels": fr.labels, "detection_scores": fr.scores, "cam_nums": fr.cam_nums}, open(tmp_path / "masks" / scene / f"{i}_data.json", "w"))
    pickle.dump(dicts, open(tmp_path / "tf" / scene, "wb"))
    # route 1: straight from the TFRecord
    converted = []
    from waymo_open_dataset.utils import frame_utils
    orig = frame_utils.convert_range_image_to_point_cloud
    monkeypatch.setattr(frame_utils, "convert_range_image_to_point_cloud", lambda fr_, *a: (converted.append(fr_.timestamp_micros), orig(fr_, *a))[1])
    f1, lanes1 = pw.load_scene(None, str(tmp_path / "masks"), scene, tfrecord=str(tmp_path / "tf" / scene))
    assert len(f1) == 3 and len(converted) == 3      # frame 2 was never converted
    # route 2: extractor -> npz -> entry point
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import extract_waymo_frames
    extract_waymo_frames.main(str(tmp_path / "tf" / scene), str(tmp_path / "frames"), scene=scene)
    f2, lanes2 = pw.load_scene(str(tmp_path / "frames"), str(tmp_path / "masks"), scene)
    assert [f.token for f in f1] == [f.token for f in f2] == [f"{scene}:{i}" for i in (0, 1, 3)]
    assert np.array_equal(lanes1, lanes2) and lanes1.shape == (6 * 301, 3)
    classes = lifting.ClassTable.waymo()
    a, b = lifting.pack_frames(f1, [lanes1], [0] * 3, classes), lifting.pack_frames(f2, [lanes2], [0] * 3, classes)
    for k in ("raw", "sweep_row_off", "sweep_xf", "frame_sweep_off", "cams", "mask_off", "mask_cam", "rle_counts", "rle_off", "class_id", "score",
              "lane", "lane_off", "pose_rt", "pose_inv"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert [(f.context_name, f.timestamp_micros) for f in f1] == [(f.context_name, f.timestamp_micros) for f in f2]
    assert a.cams.shape[1] == 5 and not a.ego_box
