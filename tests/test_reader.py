"""CPU: the native host-side loader (libcm3d_reader.so, include/cm3d_reader.h) against the Python reader that mirrors the
reference's per-frame `pickle.load` / `np.fromfile` -- same host batches, error codes instead of crashes on bad input."""
import ctypes
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
    assert len(names) == 6
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
              "class_id", "score", "lane", "lane_off", "frame_lane", "ego_xyz"):
        assert np.array_equal(getattr(x, k), getattr(y, k)), k
    assert (x.width, x.height, x.n_cams, x.raw_stride, x.max_rows_per_sweep, x.tokens, x.labels, x.ego_box) == \
           (y.width, y.height, y.n_cams, y.raw_stride, y.max_rows_per_sweep, y.tokens, y.labels, y.ego_box)


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
