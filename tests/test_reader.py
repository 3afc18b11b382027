"""CPU: the native host-side loader (libcm3d_reader.so, include/cm3d_reader.h) against the Python reader that mirrors the
reference's per-frame `pickle.load` / `np.fromfile` -- same host batches, error codes instead of crashes on bad input."""
import ctypes
import json
import os
import pickle
import re
import shutil
import subprocess

import numpy as np
import pytest

from cm3d_amd import nusc_io, pipeline_nuscenes as pn, reader, rle, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reader_library_exports_every_declared_symbol(tmp_path):
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cm3d_reader.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(cm3d_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 22 and "cm3d_reader_load_sweeps_quads" in names and "cm3d_tables_manifest" in names and "cm3d_write_results_json" in names
    h = ctypes.CDLL(reader.LIB_PATH)
    for n in names:
        assert hasattr(h, n), n
    if shutil.which("gcc"):
        c = tmp_path / "t.c"
        c.write_text('#include "cm3d_reader.h"\nint main(void) { return cm3d_reader_threads(0); }\n')
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(c)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_rle_strings_native_equals_python():
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(1, 60))
        c = rng.integers(0, 50000, n).astype(np.uint32)
        c[rng.integers(0, n)] = rng.integers(0, 2 ** 31)             # long runs: many 5-bit groups, negative deltas
        s = rle.counts_to_string(c)
        assert np.array_equal(reader.string_to_counts(s), c) and np.array_equal(rle.string_to_counts(s), c)
    assert reader.string_to_counts(b"").size == 0
    with pytest.raises(ValueError):
        reader.string_to_counts(b"0P")                                # continuation bit set on the last character


@pytest.mark.parametrize("protocol", [2, 3, 4, 5])
def test_mask_files_of_every_pickle_protocol(tmp_path, protocol):
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, i) for i in range(3)]
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"{i}_masks.pkl"
        pickle.dump(f.rles, open(p, "wb"), protocol=protocol)
        paths.append(str(p))
    paths.insert(1, None)                                             # a frame without detections has no file
    rd = reader.Reader(3, pinned=False)
    counts, rle_off, fmo, wh = rd.load_masks(paths, guess_counts=8, guess_masks=1)        # forces the grow-and-retry path
    exp = [rle.string_to_counts(r["counts"]) for f in frames for r in f.rles]
    assert np.array_equal(counts, np.concatenate(exp))
    assert np.array_equal(rle_off, np.concatenate([[0], np.cumsum([e.size for e in exp])]))
    assert list(np.diff(fmo)) == [len(frames[0].rles), 0, len(frames[1].rles), len(frames[2].rles)]
    assert np.all(wh == [cfg.width, cfg.height])


def test_bad_inputs_give_error_codes(tmp_path):
    rd = reader.Reader(2, pinned=False)
    with pytest.raises(reader.ReaderError) as e:
        rd.load_sweeps([str(tmp_path / "missing.bin")])
    assert e.value.code == reader.ERR_IO and e.value.index == 0
    odd = tmp_path / "odd.bin"
    odd.write_bytes(b"\0" * 30)                                       # not a whole number of 20-byte rows
    with pytest.raises(reader.ReaderError) as e:
        rd.load_sweeps([str(odd)])
    assert e.value.code == reader.ERR_FORMAT
    for k, obj in enumerate([{"not": "a list"}, [{"size": [4, 4], "counts": b"04"}], [{"size": [4, 4]}], [np.arange(3)]]):
        p = tmp_path / f"bad{k}.pkl"
        pickle.dump(obj, open(p, "wb"))
        with pytest.raises(reader.ReaderError) as e:                 # wrong shape / run lengths that do not cover the mask / numpy inside
            rd.load_masks([str(p)])
        assert e.value.code == reader.ERR_FORMAT and e.value.index == 0
    trunc = tmp_path / "trunc.pkl"
    trunc.write_bytes(pickle.dumps([{"size": [4, 4], "counts": b"0@"}])[:-7])
    with pytest.raises(reader.ReaderError):
        rd.load_masks([str(trunc)])


def test_native_batches_equal_the_python_reader(tmp_path):
    """prepare_scene_batch through libcm3d_reader.so (thread pool, one call per batch) = the same host batch as the
    per-frame Python reader; frames without masks drop out the same way."""
    cfg = syn.config("tiny")
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=2, frames_per_scene=3)
    os.remove(os.path.join(mask_dir, names[1], "1_masks.pkl"))        # frame without detections (missing_ok)
    pickle.dump([], open(os.path.join(mask_dir, names[0], "2_masks.pkl"), "wb"))
    import json
    json.dump({"labels": [], "detection_scores": [], "cam_nums": []}, open(os.path.join(mask_dir, names[0], "2_data.json"), "w"))
    rd = reader.Reader(4, pinned=False)
    base = ("v1.0-synth", dataroot, mask_dir, names, 3, cfg.ratio, True, None)
    a = pn.prepare_scene_batch(base)
    b = pn.prepare_scene_batch(base + (False, rd))
    assert a[0] == b[0] and len(a[0]) == 6 and len(a[1]) == len(b[1]) == 1
    x, y = a[1][0], b[1][0]
    assert x.n_frames == 4
    for k in ("raw", "sweep_row_off", "sweep_xf", "frame_sweep_off", "cams", "mask_off", "mask_cam", "mask_frame", "rle_counts", "rle_off",
              "class_id", "score", "lane", "lane_off", "frame_lane", "ego_xyz", "frame_rows"):
        assert np.array_equal(getattr(x, k), getattr(y, k), equal_nan=(k == "raw")), k      # (a frame's padding rows are NaN)
    assert (x.width, x.height, x.n_cams, x.raw_stride, x.max_rows_per_sweep, x.tokens, x.labels, x.ego_box) == \
           (y.width, y.height, y.n_cams, y.raw_stride, y.max_rows_per_sweep, y.tokens, y.labels, y.ego_box)
    # the product's layout: quads -- 12 of a row's 20 bytes leave the page cache; the native reader leaves the intensity behind
    assert x.quads and x.raw.shape[1:] == (3, 4) and y.intensity is None and x.intensity is not None


def test_integration_md_reader_binding_runs_as_written(tmp_path):
    """The reader stub printed in INTEGRATION.md section 3, executed verbatim (library path filled in; `pin_memory()` dropped where
    the process has no GPU to register pages with)."""
    import torch
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if "cm3d_reader_open" in b][0].replace('ctypes.CDLL("libcm3d_reader.so")', f'ctypes.CDLL({reader.LIB_PATH!r})')
    if not torch.cuda.is_available():
        code = code.replace(".pin_memory()", "")
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg = syn.config("tiny")
    frames = [syn.make_frame(cfg, 40 + i) for i in range(3)]
    sweep_paths, mask_paths = [], []
    for i, f in enumerate(frames):
        for k, r in enumerate(f.sweeps_raw):
            p = tmp_path / f"{i}_{k}.bin"
            np.ascontiguousarray(r, np.float32).tofile(p)
            sweep_paths.append(str(p))
        p = tmp_path / f"{i}_masks.pkl"
        pickle.dump(f.rles, open(p, "wb"))
        mask_paths.append(str(p))
    rows = sum(r.shape[0] for f in frames for r in f.sweeps_raw)
    exp = [rle.string_to_counts(r["counts"]) for f in frames for r in f.rles]
    raw, row_off, counts, rle_off, fm_off, wh = ns["load_batch"](sweep_paths, mask_paths, rows, sum(e.size for e in exp), len(exp))
    assert row_off[-1] == rows and np.array_equal(raw.numpy()[:rows], np.concatenate([r for f in frames for r in f.sweeps_raw]).astype(np.float32))
    assert np.array_equal(counts.numpy()[:rle_off[len(exp)]].view(np.uint32), np.concatenate(exp))
    assert list(fm_off) == list(np.concatenate([[0], np.cumsum([len(f.rles) for f in frames])]))
    assert (wh[:len(exp)] == [cfg.width, cfg.height]).all()


def test_more_sweep_files_than_open_file_descriptors(tmp_path):
    """A batch may name thousands of sweeps (4 scenes x 40 frames x 10 sweeps): the loader must not hold a descriptor per
    file.  1200 files under RLIMIT_NOFILE = 256."""
    import resource
    rng = np.random.default_rng(0)
    paths, rows = [], []
    for i in range(1200):
        a = rng.random((int(rng.integers(1, 6)), 5)).astype(np.float32)
        p = tmp_path / f"{i}.bin"
        a.tofile(p)
        paths.append(str(p)); rows.append(a)
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
    resource.setrlimit(resource.RLIMIT_NOFILE, (256, hard))
    try:
        rd = reader.Reader(8, pinned=False)
        raw, row_off = rd.load_sweeps(paths, 5)
    finally:
        resource.setrlimit(resource.RLIMIT_NOFILE, (soft, hard))
    assert np.array_equal(raw[:row_off[-1]], np.concatenate(rows)) and np.array_equal(np.diff(row_off), [r.shape[0] for r in rows])


def test_malformed_mask_files_are_errors_not_crashes(tmp_path):
    """A memo index out of any plausible range (BINPUT / LONG_BINPUT are file-controlled), a truncated file and an RLE string
    of 13 five-bit groups: each comes back as an error code (the Python side falls back to pickle.load or raises), never as an
    exception through the C ABI or a gigabyte allocation."""
    bad = {"memo.pkl": b"\x80\x02]r\xff\xff\xff\xff.", "trunc.pkl": pickle.dumps([{"size": [4, 4], "counts": b"04"}])[:-7],
           "notalist.pkl": pickle.dumps({"size": [4, 4]})}
    rd = reader.Reader(2, pinned=False)
    for name, blob in bad.items():
        p = tmp_path / name
        p.write_bytes(blob)
        with pytest.raises(reader.ReaderError) as e:
            rd.load_masks([str(p)])
        assert e.value.code == reader.ERR_FORMAT, name
    # 13 groups: 65 bits of payload -- a value no mask can hold, rejected; 12 groups still parse
    for groups in (12, 13, 40):
        with pytest.raises(ValueError):
            reader.string_to_counts(b"o" * groups + b"0")
    assert list(reader.string_to_counts(b"o" * 5 + b"0")) == [0x1FFFFFF]


def test_native_tables_and_manifest_equal_the_python_table_walk(tmp_path):
    """cm3d_tables_open + cm3d_tables_manifest (the table walk, <f>_data.json and the record arithmetic in native code) against
    nusc_io.NuscTables + scene_manifest + lifting.pack_manifest (Python): every array of the packed batch bit for bit, tokens,
    scene sizes, map locations; a frame without mask files under missing_ok; a label outside the class table reported, not
    swallowed."""
    from cm3d_amd import lifting
    cfg = syn.config("tiny", n_sweeps=3)
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=3, frames_per_scene=4, lane_points=300)
    os.remove(os.path.join(mask_dir, names[1], "2_masks.pkl"))          # a frame without detections has no files
    os.remove(os.path.join(mask_dir, names[1], "2_data.json"))
    rd = reader.Reader(4, pinned=False)
    nt = reader.Tables(rd, dataroot, "v1.0-synth")
    pt = nusc_io.NuscTables("v1.0-synth", dataroot, annotations=False)
    assert nt.scene_names() == [s["name"] for s in pt.t["scene"].values()]
    for n in names:
        assert nt.scene_samples(n) == 4 and nt.location(n) == pt.location(pt.scene_by_name(n))
    toks, rows = nt.job_tokens(names[::-1])
    assert toks == [s["token"] for n in names[::-1] for s in pt.samples_of_scene(pt.scene_by_name(n))]
    sample_rows = list(pt.t["sample"].keys())
    assert [sample_rows[r] for r in rows] == toks
    classes = lifting.ClassTable.nuscenes()
    man = nt.manifest(names, mask_dir, 3, cfg.ratio, classes.names, missing_ok=True)
    # the Python path on the same scenes
    pman, lanes, fl = [], [], []
    for k, n in enumerate(names):
        ms = nusc_io.scene_manifest(pt, pt.scene_by_name(n), mask_dir, n_sweeps=3, ratio=cfg.ratio, missing_ok=True)
        pman.extend(ms); lanes.append(nusc_io.load_lane_points(dataroot, pt.location(pt.scene_by_name(n)))); fl.extend([k] * len(ms))
    assert man.n_frames == len(pman) == 12
    assert [sample_rows[r] for r in man.sample_index] == [m.token for m in pman]
    assert np.array_equal(man.cams.view(np.uint32), np.stack([m.cams for m in pman]).view(np.uint32))
    assert np.array_equal(man.sweep_xf.view(np.uint32), np.concatenate([m.sweep_xf for m in pman]).astype(np.float32).view(np.uint32))
    assert np.array_equal(man.ego_xyz, np.stack([m.ego_xyz for m in pman]))
    assert np.array_equal(man.frame_sweep_off, np.concatenate([[0], np.cumsum([len(m.sweep_paths) for m in pman])]))
    assert np.array_equal(man.frame_mask_off, np.concatenate([[0], np.cumsum([len(m.labels) for m in pman])]))
    assert man.frame_mask_off[7] == man.frame_mask_off[6]                 # the frame without files: no masks
    assert np.array_equal(man.mask_cam, np.concatenate([m.cam_nums for m in pman]))
    assert np.array_equal(man.score, np.concatenate([np.asarray(m.scores, np.float64) for m in pman]))
    assert np.array_equal(man.class_id, [classes.index(lifting.get_detection_name(l)) for m in pman for l in m.labels])
    # the bulk data through the manifest's own file lists = through path lists
    raw, row_off = man.load_sweeps(5)
    raw2, row_off2 = rd.load_sweeps([p for m in pman for p in m.sweep_paths], 5)
    assert np.array_equal(row_off, row_off2) and np.array_equal(raw, raw2)
    c1, o1, f1, wh1 = man.load_masks()
    c2, o2, f2, wh2 = rd.load_masks([m.mask_path for m in pman])
    assert np.array_equal(c1, c2) and np.array_equal(o1, o2) and np.array_equal(f1, f2) and np.array_equal(wh1, wh2)
    assert np.array_equal(f1, man.frame_mask_off)
    # missing files without missing_ok, and an unknown label: errors that name the frame
    with pytest.raises(reader.ReaderError) as e:
        nt.manifest(names, mask_dir, 3, cfg.ratio, classes.names, missing_ok=False)
    assert e.value.code == reader.ERR_IO and e.value.index == 6
    d = json.load(open(os.path.join(mask_dir, names[0], "1_data.json")))
    d["labels"][0] = "unicorn"
    json.dump(d, open(os.path.join(mask_dir, names[0], "1_data.json"), "w"))
    with pytest.raises(reader.ReaderError) as e:
        nt.manifest(names, mask_dir, 3, cfg.ratio, classes.names, missing_ok=True)
    assert e.value.code == reader.ERR_FORMAT and e.value.index == 1 and "unicorn" in str(e.value)


def test_native_result_writer_equals_json_dump(tmp_path):
    """cm3d_write_results_json against lifting.nuscenes_results_json (itself held to json.dumps of the reference's dicts in
    tests/test_host_logic.py): byte for byte, on records whose floats cover every branch of Python's repr -- integers, tiny and
    huge magnitudes (exponent notation from 1e16 and below 1e-4), negative zero, subnormals, non-finite values, 17-digit
    mantissas -- and samples without boxes."""
    from cm3d_amd import lifting
    rng = np.random.default_rng(5)
    classes = lifting.ClassTable.nuscenes()
    tokens = [f"tok-{i:04d}" for i in range(40)] + ['quote"and\\\\slash', "ünï"]
    special = [0.0, -0.0, 1.0, -1.0, 100.0, 1e15, 1e16, 1.5e16, 123456789012345680.0, 1e22, 1e-4, 9.999e-5, 1e-5, 1.2345e-7, 5e-324, 2.2250738585072014e-308,
               1.7976931348623157e308, 0.1, 1 / 3, 2 / 3, 1e23, 1234.5, 600.1234567890123, 1600.987654321, float("inf"), float("-inf"), float("nan"),
               4.35, 0.30000000000000004, 123456.78901234567, 9007199254740993.0, 0.001, 0.0001, 12345678.0, 1e-10, 3.14e+100]
    n = 700
    rec = np.zeros((n, 10))
    rec[:, 0:3] = rng.normal(scale=[800, 800, 2], size=(n, 3)) + [600, 1600, 0]
    rec[:, 3:5] = rng.normal(size=(n, 2))
    rec[:, 5] = rng.integers(0, len(tokens), n)
    rec[rec[:, 5] == 7, 5] = 8                                           # sample 7 stays without boxes
    rec[:, 7] = np.round(rng.uniform(0.3, 1, n), 2)
    rec[:, 8] = rng.integers(0, len(classes.names), n)
    vals = rng.choice(special, size=(n, 6))
    use = rng.random((n, 6)) < 0.3
    for k, col in enumerate((0, 1, 2, 3, 4, 7)):
        rec[use[:, k], col] = vals[use[:, k], k]
    meta = {"use_camera": True, "use_lidar": False, "use_radar": False, "use_map": True, "use_external": False}
    want, n_boxes = lifting.nuscenes_results_json(rec, tokens, classes, meta)
    got = lifting.nuscenes_results_json_native(rec, tokens, classes, meta)
    assert n_boxes == n and got == want.encode()
    assert lifting.nuscenes_results_json_native(np.zeros((0, 10)), tokens[:3], classes, meta) == lifting.nuscenes_results_json(np.zeros((0, 10)), tokens[:3], classes, meta)[0].encode()


def test_native_result_writer_in_parts_equals_the_whole(tmp_path):
    """The entry point formats every batch's share of the result file while later batches are still on the GPU: the parts, joined,
    are the file the one-call writer produces."""
    from cm3d_amd import lifting
    rng = np.random.default_rng(6)
    classes = lifting.ClassTable.nuscenes()
    tokens = [f"tok-{i:04d}" for i in range(50)]
    rec = np.zeros((400, 10))
    rec[:, 0:3] = rng.normal(scale=50, size=(400, 3))
    rec[:, 3:5] = rng.normal(size=(400, 2))
    rec[:, 5] = np.sort(rng.integers(0, 50, 400))
    rec[rec[:, 5] == 20, 5] = 21
    rec[:, 7] = np.round(rng.uniform(0.3, 1, 400), 2)
    rec[:, 8] = rng.integers(0, len(classes.names), 400)
    meta = {"use_camera": True}
    whole = lifting.nuscenes_results_json_native(rec, tokens, classes, meta)
    parts = []
    for first, count in ((0, 17), (17, 8), (25, 0), (25, 25)):
        sel = (rec[:, 5] >= first) & (rec[:, 5] < first + count)
        p = lifting.nuscenes_results_json_native(rec[sel], tokens, classes, meta, part=(first, count))
        if count:
            parts.append(p)
    assert lifting.nuscenes_results_json_head(meta) + b", ".join(parts) + b"}}" == whole


def test_integration_md_tables_binding_runs_as_written(tmp_path):
    """The tables / manifest stub printed in INTEGRATION.md section 3, executed verbatim (library path filled in), against the
    Python table walk."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if "cm3d_tables_manifest" in b][0].replace('ctypes.CDLL("libcm3d_reader.so")', f'ctypes.CDLL({reader.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg = syn.config("tiny", n_sweeps=3)
    dataroot, mask_dir, names = nusc_io.write_synthetic_dataset(str(tmp_path), cfg, n_scenes=2, frames_per_scene=3, lane_points=200)
    got = ns["scene_inputs"](dataroot, "v1.0-synth", names[1], mask_dir, 3, cfg.ratio)
    pt = nusc_io.NuscTables("v1.0-synth", dataroot, annotations=False)
    pman = nusc_io.scene_manifest(pt, pt.scene_by_name(names[1]), mask_dir, n_sweeps=3, ratio=cfg.ratio)
    assert np.array_equal(got["cams"].view(np.uint32), np.stack([m.cams for m in pman]).view(np.uint32))
    assert np.array_equal(got["sweep_xf"].view(np.uint32), np.concatenate([m.sweep_xf for m in pman]).astype(np.float32).view(np.uint32))
    assert np.array_equal(got["score"], np.concatenate([np.asarray(m.scores, np.float64) for m in pman]))
    assert np.array_equal(got["mask_cam"], np.concatenate([m.cam_nums for m in pman])) and got["class_id"].size == got["score"].size


def test_native_quads_equal_the_numpy_packing(tmp_path):
    """cm3d_reader_load_sweeps_quads (files -> quad layout, frames padded to whole quads) against lifting.rows_to_quads on the rows
    the plain loader returns: ragged sweeps (a sweep that starts inside a quad), an empty sweep, a frame of one row, both
    intensity settings -- and the rows layout of the same files stays what it was."""
    from cm3d_amd import lifting
    rng = np.random.default_rng(11)
    sizes = [7, 0, 13, 1, 256, 5, 4, 3]
    fso = np.array([0, 3, 4, 6, 8], np.int32)                   # frames of 3, 1, 2, 2 sweeps
    paths = []
    for i, n in enumerate(sizes):
        a = rng.normal(0, 30, (n, 5)).astype(np.float32)
        p = tmp_path / f"s{i}.bin"
        a.tofile(p)
        paths.append(str(p))
    rd = reader.Reader(3, pinned=False)
    rows, off = rd.load_sweeps(paths, 5)
    want_q, want_i, want_off, want_rows = lifting.rows_to_quads(np.asarray(rows), off, fso)
    for intensity in (False, True):
        q, it, o, fr = rd.load_sweeps_quads(paths, fso, 5, intensity)
        assert np.array_equal(np.asarray(q), want_q, equal_nan=True) and np.array_equal(o, want_off) and np.array_equal(fr, want_rows)
        assert (it is None) == (not intensity)
        if intensity:
            assert np.array_equal(np.asarray(it), want_i)
    assert int(want_off[-1]) % 4 == 0 and np.all(want_off[fso[:-1]] % 4 == 0) and want_rows.tolist() == [20, 1, 261, 7]
    with pytest.raises(reader.ReaderError):
        rd.load_sweeps_quads(paths, np.array([0, 3, 9], np.int32), 5)          # frames that do not cover the files
