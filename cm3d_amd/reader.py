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
        h.cm3d_reader_load_masks.restype, h.cm3d_reader_load_masks.argtypes = C.c_int, [p, p, i32, p, i64, p, p, p, i32, p, p]
        h.cm3d_rle_string_to_counts.restype, h.cm3d_rle_string_to_counts.argtypes = i64, [C.c_char_p, i64, p, i64]
        _lib = h
    return _lib


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
