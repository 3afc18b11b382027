"""Loader of libcm3d_hip.so (the C-ABI declared in include/cm3d_hip.h).

There is no CPU fallback: if the shared library is missing or a symbol cannot
be resolved this raises, and every product entry point fails with it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM3D_LIB") or os.path.join(_HERE, "libcm3d_hip.so")      # CM3D_LIB: experiments only

ABI_VERSION = 4
CAM_STRIDE = 64
SWEEP_XF_STRIDE = 24
MAX_CAMS = 8
MAX_MASKS_PER_FRAME = 1024
BOX_STRIDE = 10
MEDOID_TILE = 64
STATUS_WORDS = 4
BBOX_STRIDE = 8          # int32 per mask in `bbox`: eroded bounds [0..3], stored rectangle xw0, y0, wc, rows [4..7] (include/cm3d_hip.h)
MAX_MATCH_BOXES = 1024
MAX_FUSED_SWEEPS = 16
MATCH_BOX_STRIDE = 6
RAW_QUADS = 3            # raw_stride value of the quad layout (include/cm3d_hip.h, cm3d_sweep_prep)

_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/cm3d_hip.h one to one
SIGNATURES = {
    "cm3d_abi_version": (_i32, []),
    "cm3d_project_workgroups_per_cu": (_i32, [_i32]),
    "cm3d_error_string": (C.c_char_p, [_i32]),
    "cm3d_removed_words": (_i64, [_i32, _i32]),
    "cm3d_batch_begin": (_i32, [_p, _p, _i32, _p, _i64, _p]),
    "cm3d_sweep_prep": (_i32, [_p, _i32, _p, _p, _i32, _i32, _p, _p, _i32, _f32, _p, _i32, _p, _p, _p, _p]),
    "cm3d_rle_workspace_bytes": (_i64, [_i32]),
    "cm3d_rle_to_dense": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _i64, _p]),
    "cm3d_erode_pack": (_i32, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "cm3d_rle_erode_pack": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _p]),
    "cm3d_rle_erode_pack_begin": (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _p, _p, _i32, _p, _i64, _p]),
    "cm3d_project_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cm3d_project_hit_rows": (_i32, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "cm3d_project_hits": (_i32, [_p, _p, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _i32,
                                 _p, _p, _p, _p, _i64, _p, _p, _p]),
    "cm3d_sweep_project_hits": (_i32, [_p, _i32, _p, _p, _i32, _i32, _p, _p, _f32, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _i32, _p, _p,
                                       _p, _p, _i32, _i32, _i32, _f32, _i32, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "cm3d_compact_hits": (_i32, [_p, _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p,
                                 _i64, _p]),
    "cm3d_tile_work_bytes": (_i64, [_i32, _i32]),
    "cm3d_selftest_sqrt": (_i32, [C.c_uint32, C.c_uint32, _p, _p, _p]),
    "cm3d_selftest_div": (_i32, [C.c_uint64, C.c_uint64, _p, _p]),
    "cm3d_selftest_mfma": (_i32, [C.c_uint64, _i32, _p, _p]),
    "cm3d_medoid_workspace_bytes": (_i64, [_i32, _i32]),
    "cm3d_medoid": (_i32, [_p, _p, _p, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    "cm3d_medoid2": (_i32, [_p, _p, _p, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "cm3d_lane_grid_bytes": (_i64, [_i32, _i32]),
    "cm3d_lane_grid_build": (_i32, [_p, _p, _i32, _i32, _p, _i64, _p]),
    "cm3d_lane_nn_workspace_bytes": (_i64, [_i32]),
    "cm3d_lane_nn": (_i32, [_p, _p, _p, _i32, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "cm3d_circle_nms": (_i32, [_p, _p, _p, _p, _p, _i32, _p, _i32, _p, _p]),
    "cm3d_box_nms": (_i32, [_p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "cm3d_centroid_transform": (_i32, [_p, _p, _p, _i32, _p, _p, _p]),
    "cm3d_bev_match_workspace_bytes": (_i64, [_i64]),
    "cm3d_bev_match": (_i32, [_p, _p, _i32, _p, _p, _i32, _p, _i32, _i64, C.c_double, _p, _p, _p, _p, _p, _i64, _p]),
}


class Cm3dError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (loads on first use).  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Cm3dError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the lifting path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.cm3d_abi_version() != ABI_VERSION:
            raise Cm3dError("libcm3d_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().cm3d_error_string(code).decode()
        raise Cm3dError(f"{what}: {msg} ({code})")
