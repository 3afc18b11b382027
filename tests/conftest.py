import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


# The raw rows of a batch exist in two layouts (include/cm3d_hip.h, cm3d_sweep_prep): the files' own rows and the quad layout
# the product packs by default.  Every GPU parity test that packs frames runs once per layout.
_LAYOUT_MODULES = {"test_gpu_parity", "test_gpu_magnitude", "test_gpu_campaign"}
_LAYOUT_GOLDEN_PREFIXES = ("test_g2_", "test_g2b_", "test_g2d_", "test_g2e_", "test_g7r_", "test_integration_md")


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    name = metafunc.function.__name__
    if mod in _LAYOUT_MODULES or (mod == "test_gpu_golden" and name.startswith(_LAYOUT_GOLDEN_PREFIXES)):
        if "raw_layout" in metafunc.fixturenames:
            metafunc.parametrize("raw_layout", ["quads", "rows"], indirect=True)


@pytest.fixture(autouse=True)
def raw_layout(request, monkeypatch):
    layout = getattr(request, "param", None)
    if layout is not None:
        monkeypatch.setenv("CM3D_RAW_LAYOUT", layout)
    return layout
