"""ctypes binding of libcm3d_reader.so (include/cm3d_reader.h): the native host-side loader.

Replaces, for a whole batch of frames at once and on a pool of threads, the reference's per-frame
`pickle.load` + pycocotools decode (src/nuscenes/2d_to_3d.py:422-428) and `np.fromfile` of every sweep
(:437-441, utils/pcd.py:246-257).  Outputs land in staging buffers that are page-locked when a GPU is present, laid
out like the kernels' inputs, so `LiftEngine.upload` can copy them asynchronously.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcm3d_reader.so")
OK, ERR_ARG, ERR_IO, ERR_FORMAT, ERR_CAPACITY = 0, -1, -2, -3, -4


class ReaderError(RuntimeError):
    def __init__(self, code, what, index=-1):
        super().__init__(f"{what}: code {code}" + (f" (file {index})" if index >= 0 else ""))
        self.code, self.index = code, index


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ReaderError(ERR_ARG, f"{LIB_PATH} is missing: build it with `make -C cm3d_amd/csrc`")
        h = C.CDLL(LIB_PATH)
        p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        h.cm3d_reader_open.restype, h.cm3d_reader_open.argtypes = p, [i32]
        h.cm3d_reader_close.restype, h.cm3d_reader_close.argtypes = None, [p]
        h.cm3d_reader_threads.restype, h.cm3d_reader_threads.argtypes = i32, [p]
        h.cm3d_reader_load_sweeps.restype, h.cm3d_reader_load_sweeps.argtypes = C.c_int, [p, p, i32, i32, p, i64, p, p]
        h.cm3d_reader_load_sweeps_quads.restype, h.cm3d_reader_load_sweeps_quads.argtypes = C.c_int, [p, p, i32, i32, p, i32, p, p, i64, p, p, p]
        h.cm3d_reader_load_masks.restype, h.cm3d_reader_load_masks.argtypes = C.c_int, [p, p, i32, p, i64, p, p, p, i32, p, p]
        h.cm3d_rle_string_to_counts.restype, h.cm3d_rle_string_to_counts.argtypes = i64, [C.c_char_p, i64, p, i64]
        # tables / manifests / writer
        h.cm3d_tables_open.restype, h.cm3d_tables_open.argtypes = p, [p, C.c_char_p, C.c_char_p, p]
        h.cm3d_tables_close.restype, h.cm3d_tables_close.argtypes = None, [p]
        h.cm3d_tables_scene_samples.restype, h.cm3d_tables_scene_samples.argtypes = i32, [p, C.c_char_p]
        h.cm3d_tables_scene_location.restype, h.cm3d_tables_scene_location.argtypes = i32, [p, C.c_char_p, p, i32]
        h.cm3d_tables_scene_names.restype, h.cm3d_tables_scene_names.argtypes = i64, [p, p, i64]
        h.cm3d_tables_job_tokens.restype, h.cm3d_tables_job_tokens.argtypes = i64, [p, p, i32, p, i64, p, i64]
        h.cm3d_tables_manifest.restype = p
        h.cm3d_tables_manifest.argtypes = [p, p, p, i32, C.c_char_p, i32, C.c_double, p, i32, i32, p]
        h.cm3d_manifest_close.restype, h.cm3d_manifest_close.argtypes = None, [p]
        h.cm3d_manifest_sizes.restype, h.cm3d_manifest_sizes.argtypes = None, [p, p]
        h.cm3d_manifest_bad_label.restype, h.cm3d_manifest_bad_label.argtypes = C.c_char_p, [p]
        h.cm3d_manifest_copy.restype, h.cm3d_manifest_copy.argtypes = C.c_int, [p] + [p] * 13
        h.cm3d_manifest_load_sweeps.restype, h.cm3d_manifest_load_sweeps.argtypes = C.c_int, [p, p, i32, p, i64, p, p]
        h.cm3d_manifest_load_sweeps_quads.restype, h.cm3d_manifest_load_sweeps_quads.argtypes = C.c_int, [p, p, i32, p, p, i64, p, p, p]
        h.cm3d_manifest_load_masks.restype, h.cm3d_manifest_load_masks.argtypes = C.c_int, [p, p, p, i64, p, p, p, i32, p, p]
        h.cm3d_write_results_json.restype = i64
        h.cm3d_write_results_json.argtypes = [p, i64, C.c_char_p, i32, p, p, p, i32, C.c_char_p, p, i64]
        _lib = h
    return _lib


def usable_cores():
    """Cores this process may use: the scheduler affinity, cut by the cgroup CPU quota when there is one (a container that sees
    256 cores and is granted 16 must not start 64 reader threads: measured 8 k against 15 k frames/s end to end)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _paths(paths):
    arr = (C.c_char_p * len(paths))()
    arr[:] = [None if p is None else os.fsencode(p) for p in paths]
    return arr


def _staging(n, dtype, pinned):
    """1-D staging buffer of n elements: page-locked when asked for (torch owns the allocation), else plain numpy."""
    if pinned:
        import torch
        t = torch.empty(max(int(n), 1), dtype={np.float32: torch.float32, np.uint32: torch.int32, np.int32: torch.int32}[dtype],
                        pin_memory=True)
        return t.numpy().view(dtype), t
    return np.empty(max(int(n), 1), dtype), None


class Reader:
    """A pool of reader threads plus growable staging buffers.  One instance per consumer thread."""

    def __init__(self, threads=0, pinned=None):
        if int(threads) <= 0:
            threads = min(64, usable_cores())          # 0 = one thread per core this process may use
        self.h = lib().cm3d_reader_open(int(threads))
        if not self.h:
            raise ReaderError(ERR_ARG, "cm3d_reader_open")
        if pinned is None:
            try:
                import torch
                pinned = torch.cuda.is_available()
            except ImportError:
                pinned = False
        self.pinned = bool(pinned)
        self._keep = []            # torch owners of the page-locked buffers handed out

    @property
    def threads(self):
        return int(lib().cm3d_reader_threads(self.h))

    def close(self):
        if self.h:
            lib().cm3d_reader_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_sweeps(self, paths, stride=5, alloc=None):
        """-> (raw (rows, stride) float32, sweep_row_off (n+1,) int32); the rows of all files back to back.
        alloc(n_floats) -> writable float32 array: where the rows go instead of this reader's own staging buffer (a reader
        process hands in a shared-memory segment, so the rows are copied once: page cache -> segment)."""
        L, n = lib(), len(paths)
        arr = _paths(paths)
        off = np.zeros(n + 1, np.int32)
        bad = C.c_int32(-1)
        rc = L.cm3d_reader_load_sweeps(self.h, arr, n, stride, None, 0, off.ctypes.data, C.byref(bad))
        if rc not in (OK, ERR_CAPACITY):
            raise ReaderError(rc, "cm3d_reader_load_sweeps", bad.value)
        rows = int(off[-1])
        if alloc is not None:
            raw, owner = np.asarray(alloc(max(rows * stride, 1))).reshape(-1), None
        else:
            raw, owner = _staging(rows * stride, np.float32, self.pinned)
        if rows:
            rc = L.cm3d_reader_load_sweeps(self.h, arr, n, stride, raw.ctypes.data, rows, off.ctypes.data, C.byref(bad))
            if rc != OK:
                raise ReaderError(rc, "cm3d_reader_load_sweeps", bad.value)
        out = raw[:rows * stride].reshape(rows, stride)
        if owner is not None:
            self._keep = [owner] + self._keep[:1]          # the numpy view does not own pinned memory: keep the last few alive
            # (few: a released buffer goes back to torch's page-locked cache, and the next batch takes it from there instead of
            # pinning 200 MB afresh)
            out = _Owned(out, owner)
        return out, off

    def load_sweeps_quads(self, paths, frame_sweep_off, file_stride=5, intensity=False, alloc=None):
        """The same files straight into the quad layout of include/cm3d_hip.h (x, y, z of four rows side by side, frames padded to
        whole quads): -> (quads (R/4, 3, 4) float32, intensity (R,) float32 or None, sweep_row_off (n+1,) int32 in the padded
        numbering, frame_rows (F,) int32).  12 of a row's bytes reach the batch (16 with intensity=True) instead of 4 * file_stride.
        alloc(n_floats) as in load_sweeps (the quads only)."""
        L, n = lib(), len(paths)
        arr = _paths(paths)
        fso = np.ascontiguousarray(frame_sweep_off, np.int32)
        F = len(fso) - 1
        off, frows = np.zeros(n + 1, np.int32), np.zeros(max(F, 1), np.int32)
        bad = C.c_int32(-1)
        rc = L.cm3d_reader_load_sweeps_quads(self.h, arr, n, file_stride, fso.ctypes.data, F, None, None, 0, off.ctypes.data, frows.ctypes.data, C.byref(bad))
        if rc not in (OK, ERR_CAPACITY):
            raise ReaderError(rc, "cm3d_reader_load_sweeps_quads", bad.value)
        rows = int(off[-1])
        if alloc is not None:
            raw, owner = np.asarray(alloc(max(rows * 3, 4))).reshape(-1), None
        else:
            raw, owner = _staging(max(rows * 3, 4), np.float32, self.pinned)
        inten, iowner = (_staging(max(rows, 1), np.float32, self.pinned) if intensity else (None, None))
        if rows:
            rc = L.cm3d_reader_load_sweeps_quads(self.h, arr, n, file_stride, fso.ctypes.data, F, raw.ctypes.data, inten.ctypes.data if intensity else None,
                                                 rows, off.ctypes.data, frows.ctypes.data, C.byref(bad))
            if rc != OK:
                raise ReaderError(rc, "cm3d_reader_load_sweeps_quads", bad.value)
        out = raw[:rows * 3].reshape(rows // 4, 3, 4)
        iout = inten[:rows] if intensity else None
        if owner is not None:
            self._keep = [owner, iowner] + self._keep[:2]
            out = _Owned(out, owner)
            if iout is not None:
                iout = _Owned(iout, iowner)
        return out, iout, off, frows[:F]

    def load_masks(self, paths, guess_counts=1 << 20, guess_masks=1 << 12):
        """-> (rle_counts uint32, rle_off int32 (M+1), frame_mask_off int32 (n+1), mask_wh int32 (M,2)).
        paths[i] None or '' = frame i has no mask file."""
        L, n = lib(), len(paths)
        arr = _paths([p or None for p in paths])
        fmo = np.zeros(n + 1, np.int32)
        need = np.zeros(2, np.int64)
        bad = C.c_int32(-1)
        cap_c, cap_m = int(guess_counts), int(guess_masks)
        for _ in range(2):
            counts, owner = _staging(cap_c, np.uint32, self.pinned)
            rle_off = np.zeros(cap_m + 1, np.int32)
            wh = np.zeros((max(cap_m, 1), 2), np.int32)
            rc = L.cm3d_reader_load_masks(self.h, arr, n, counts.ctypes.data, cap_c, rle_off.ctypes.data, fmo.ctypes.data, wh.ctypes.data,
                                          cap_m, need.ctypes.data, C.byref(bad))
            if rc == OK:
                nc, nm = int(need[0]), int(need[1])
                c = counts[:nc]
                if owner is not None:
                    self._keep = [owner] + self._keep[:1]
                    c = _Owned(c, owner)
                return c, rle_off[:nm + 1].copy(), fmo, wh[:nm].copy()
            if rc != ERR_CAPACITY:
                raise ReaderError(rc, "cm3d_reader_load_masks", bad.value)
            cap_c, cap_m = int(need[0]) + 16, int(need[1]) + 1
        raise ReaderError(rc, "cm3d_reader_load_masks")


class _Owned(np.ndarray):
    """numpy view of a page-locked torch tensor that keeps its owner alive."""

    def __new__(cls, arr, owner):
        obj = np.asarray(arr).view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def string_to_counts(s: bytes) -> np.ndarray:
    """COCO compressed string -> uint32 run lengths through the native parser (same result as cm3d_amd.rle.string_to_counts)."""
    L = lib()
    n = L.cm3d_rle_string_to_counts(s, len(s), None, 0)
    if n < 0:
        raise ValueError("malformed RLE string")
    out = np.empty(max(int(n), 1), np.uint32)
    if L.cm3d_rle_string_to_counts(s, len(s), out.ctypes.data, n) != n:
        raise ValueError("malformed RLE string")
    return out[:n]


class Tables:
    """The nuScenes tables in native memory (cm3d_tables_open): what NuScenes(VER_NAME, INPUT_PATH) holds for the frame loop of the
    reference (src/nuscenes/2d_to_3d.py:382, :415-503), parsed once by the reader's thread pool."""

    def __init__(self, rd: Reader, dataroot, version):
        err = C.c_int32(0)
        self.rd, self.dataroot, self.version = rd, dataroot, version
        self.h = lib().cm3d_tables_open(rd.h, os.fsencode(dataroot), os.fsencode(version), C.byref(err))
        if not self.h:
            raise ReaderError(err.value, f"cm3d_tables_open({dataroot!r}, {version!r})")

    def close(self):
        if self.h:
            lib().cm3d_tables_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scene_names(self):
        L = lib()
        n = L.cm3d_tables_scene_names(self.h, None, 0)
        buf = C.create_string_buffer(int(n))
        L.cm3d_tables_scene_names(self.h, buf, n)
        return [x.decode() for x in buf.raw[:n].split(b"\0")[:-1]]

    def scene_samples(self, name):
        n = lib().cm3d_tables_scene_samples(self.h, name.encode())
        if n < 0:
            raise KeyError(name)
        return int(n)

    def location(self, name):
        buf = C.create_string_buffer(256)
        if lib().cm3d_tables_scene_location(self.h, name.encode(), buf, 256) < 0:
            raise KeyError(name)
        return buf.value.decode()

    def job_tokens(self, names):
        """-> (sample tokens of the scenes in job order, their rows in sample.json as an int32 array)."""
        L = lib()
        arr = _paths(names)
        need = L.cm3d_tables_job_tokens(self.h, arr, len(names), None, 0, None, 0)
        if need < 0:
            raise KeyError(names)
        n = sum(self.scene_samples(s) for s in names)
        buf = C.create_string_buffer(int(need))
        rows = np.zeros(max(n, 1), np.int32)
        L.cm3d_tables_job_tokens(self.h, arr, len(names), buf, need, rows.ctypes.data, n)
        return [x.decode() for x in buf.raw[:need].split(b"\0")[:-1]], rows[:n]

    def manifest(self, names, mask_dir, n_sweeps, ratio, class_names, missing_ok=False):
        return Manifest(self, names, mask_dir, n_sweeps, ratio, class_names, missing_ok)


class Manifest:
    """The table walk of a batch of scenes done natively (cm3d_tables_manifest; reference :415-441, :489-503, :423): the
    per-frame arrays of a lift batch -- sweep transforms, camera records, ego positions, labels as class ids, scores, camera
    numbers -- plus the sweep and mask files, which `load` reads through the same handle without a path list ever crossing into
    Python."""

    def __init__(self, tables, names, mask_dir, n_sweeps, ratio, class_names, missing_ok):
        L = lib()
        err = C.c_int32(0)
        self.tables = tables
        self.h = L.cm3d_tables_manifest(tables.h, tables.rd.h, _paths(names), len(names), os.fsencode(mask_dir), int(n_sweeps), float(ratio),
                                        _paths(class_names), len(class_names), 1 if missing_ok else 0, C.byref(err))
        if not self.h:
            raise ReaderError(err.value, "cm3d_tables_manifest")
        sz = np.zeros(6, np.int64)
        L.cm3d_manifest_sizes(self.h, sz.ctypes.data)
        if sz[5] >= 0:
            label = L.cm3d_manifest_bad_label(self.h).decode()
            code = err.value
            self.close()
            raise ReaderError(code, f"cm3d_tables_manifest: {label!r}", int(sz[5]))
        F, S, M = int(sz[0]), int(sz[1]), int(sz[2])
        self.n_frames, self.n_sweeps, self.n_masks = F, S, M
        self.sample_index = np.zeros(F, np.int32)
        self.frame_sweep_off = np.zeros(F + 1, np.int32)
        self.sweep_xf = np.zeros((S, 24), np.float32)
        self.cams = np.zeros((F, 6, 64), np.float32)
        self.ego_xyz = np.zeros((F, 3), np.float64)
        self.frame_mask_off = np.zeros(F + 1, np.int32)
        self.mask_cam = np.zeros(M, np.int32)
        self.class_id = np.zeros(M, np.int32)
        self.score = np.zeros(M, np.float64)
        ptr = lambda a: a.ctypes.data if a.size else None
        rc = L.cm3d_manifest_copy(self.h, ptr(self.sample_index), ptr(self.frame_sweep_off), ptr(self.sweep_xf), ptr(self.cams), ptr(self.ego_xyz),
                                  ptr(self.frame_mask_off), ptr(self.mask_cam), ptr(self.class_id), ptr(self.score), None, None, None, None)
        if rc != OK:
            raise ReaderError(rc, "cm3d_manifest_copy")

    def close(self):
        if self.h:
            lib().cm3d_manifest_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_sweeps(self, stride=5, rd=None):
        """-> (raw (rows, stride) float32 in a page-locked staging buffer, sweep_row_off int32).  rd: the Reader whose threads and
        staging buffers to use (a Reader serves one calling thread at a time: a caller that reads the sweeps on another thread
        than the one that walks the tables brings a second one)."""
        L, rd = lib(), (rd or self.tables.rd)
        off = np.zeros(self.n_sweeps + 1, np.int32)
        bad = C.c_int32(-1)
        rc = L.cm3d_manifest_load_sweeps(rd.h, self.h, stride, None, 0, off.ctypes.data, C.byref(bad))
        if rc not in (OK, ERR_CAPACITY):
            raise ReaderError(rc, "cm3d_manifest_load_sweeps", bad.value)
        rows = int(off[-1])
        raw, owner = _staging(rows * stride, np.float32, rd.pinned)
        if rows:
            rc = L.cm3d_manifest_load_sweeps(rd.h, self.h, stride, raw.ctypes.data, rows, off.ctypes.data, C.byref(bad))
            if rc != OK:
                raise ReaderError(rc, "cm3d_manifest_load_sweeps", bad.value)
        out = raw[:rows * stride].reshape(rows, stride)
        if owner is not None:
            rd._keep = [owner] + rd._keep[:1]
            out = _Owned(out, owner)
        return out, off

    def load_sweeps_quads(self, file_stride=5, rd=None, intensity=False):
        """load_sweeps into the quad layout (Reader.load_sweeps_quads): -> (quads, intensity or None, sweep_row_off, frame_rows)."""
        L, rd = lib(), (rd or self.tables.rd)
        F = self.n_frames
        off, frows = np.zeros(self.n_sweeps + 1, np.int32), np.zeros(max(F, 1), np.int32)
        bad = C.c_int32(-1)
        rc = L.cm3d_manifest_load_sweeps_quads(rd.h, self.h, file_stride, None, None, 0, off.ctypes.data, frows.ctypes.data, C.byref(bad))
        if rc not in (OK, ERR_CAPACITY):
            raise ReaderError(rc, "cm3d_manifest_load_sweeps_quads", bad.value)
        rows = int(off[-1])
        raw, owner = _staging(max(rows * 3, 4), np.float32, rd.pinned)
        inten, iowner = (_staging(max(rows, 1), np.float32, rd.pinned) if intensity else (None, None))
        if rows:
            rc = L.cm3d_manifest_load_sweeps_quads(rd.h, self.h, file_stride, raw.ctypes.data, inten.ctypes.data if intensity else None, rows,
                                                   off.ctypes.data, frows.ctypes.data, C.byref(bad))
            if rc != OK:
                raise ReaderError(rc, "cm3d_manifest_load_sweeps_quads", bad.value)
        out = raw[:rows * 3].reshape(rows // 4, 3, 4)
        iout = inten[:rows] if intensity else None
        if owner is not None:
            rd._keep = [owner, iowner] + rd._keep[:2]
            out = _Owned(out, owner)
            if iout is not None:
                iout = _Owned(iout, iowner)
        return out, iout, off, frows[:F]

    def load_masks(self, guess_counts=1 << 20):
        """-> (rle_counts uint32, rle_off int32 (M+1), frame_mask_off int32 (F+1), mask_wh int32 (M,2))"""
        L, rd = lib(), self.tables.rd
        fmo = np.zeros(self.n_frames + 1, np.int32)
        need = np.zeros(2, np.int64)
        bad = C.c_int32(-1)
        cap_c, cap_m = int(guess_counts), max(self.n_masks, 1)
        for _ in range(2):
            counts, owner = _staging(cap_c, np.uint32, rd.pinned)
            rle_off = np.zeros(cap_m + 1, np.int32)
            wh = np.zeros((cap_m, 2), np.int32)
            rc = L.cm3d_manifest_load_masks(rd.h, self.h, counts.ctypes.data, cap_c, rle_off.ctypes.data, fmo.ctypes.data, wh.ctypes.data, cap_m,
                                            need.ctypes.data, C.byref(bad))
            if rc == OK:
                nc, nm = int(need[0]), int(need[1])
                c = counts[:nc]
                if owner is not None:
                    rd._keep = [owner] + rd._keep[:1]
                    c = _Owned(c, owner)
                return c, rle_off[:nm + 1].copy(), fmo, wh[:nm].copy()
            if rc != ERR_CAPACITY:
                raise ReaderError(rc, "cm3d_manifest_load_masks", bad.value)
            cap_c, cap_m = int(need[0]) + 16, int(need[1]) + 1
        raise ReaderError(rc, "cm3d_manifest_load_masks")


def write_results_json(rec, tokens_json, cls_mid, cls_score, cls_tail, prefix):
    """The text of the reference's result file (json.dump of the dict of box lists, :929-930) from the gathered box records,
    written natively (cm3d_write_results_json); see lifting.nuscenes_results_json for the Python form it must equal."""
    L = lib()
    rec = np.ascontiguousarray(rec, np.float64).reshape(-1, 10)
    blob = b"".join(t.encode() + b"\0" for t in tokens_json)
    a, b, c = _paths(cls_mid), _paths(cls_score), _paths(cls_tail)
    cap = int(rec.shape[0]) * 360 + len(blob) * 2 + len(prefix or "") + 256
    for _ in range(2):
        out = C.create_string_buffer(cap)
        n = L.cm3d_write_results_json(rec.ctypes.data if rec.size else None, rec.shape[0], blob, len(tokens_json), a, b, c, len(cls_mid),
                                      None if prefix is None else prefix.encode(), out, cap)
        if n > 0:
            return out.raw[:n]
        if n == 0:
            if prefix is None and not tokens_json:
                return b""
            raise ReaderError(ERR_ARG, "cm3d_write_results_json")
        cap = int(-n) + 16
    raise ReaderError(ERR_CAPACITY, "cm3d_write_results_json")
