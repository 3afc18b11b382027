"""CPU: the C-ABI library loads and exports every symbol include/cm3d_hip.h declares
(no compute calls without a GPU), and argument validation returns error codes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cm3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cm3d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from cm3d_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/cm3d_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "python signatures out of sync with the header"


def test_abi_version_and_error_strings():
    from cm3d_amd import _lib
    L = _lib.lib()
    assert L.cm3d_abi_version() == _lib.ABI_VERSION
    assert L.cm3d_error_string(0) == b"ok"
    assert b"workspace" in L.cm3d_error_string(-3)


def test_argument_validation_never_launches():
    from cm3d_amd import _lib
    L = _lib.lib()
    assert L.cm3d_erode_pack(0, 1, 64, 64, 0, 0, 0) == -1                 # null pointers
    assert L.cm3d_batch_begin(0, 0, 0, 0, 0, 0) == -1
    assert L.cm3d_rle_workspace_bytes(100) == 400 + 1600
    assert L.cm3d_medoid_workspace_bytes(10, 1000) > 0 and L.cm3d_lane_nn_workspace_bytes(10) > 0 and L.cm3d_lane_grid_bytes(1, 1000) > 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    from cm3d_amd import lifting, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.Cm3dError):
        lifting.LiftEngine("cuda:0")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cm3d_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_header_is_plain_c(tmp_path):
    """include/cm3d_hip.h is the C-ABI: it must compile as C99 without any HIP / C++ / torch type."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "cm3d_hip.h"\nint main(void) { return cm3d_abi_version() > 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
