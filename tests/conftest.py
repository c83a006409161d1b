import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest` on a box without a GPU (or without the built HIP library) skips the gpu-marked tests
    instead of failing them; `-m gpu` on the GPU box selects them as before."""
    reason = None
    try:
        import torch
        if not torch.cuda.is_available():
            reason = "no HIP GPU visible"
    except Exception as e:  # pragma: no cover
        reason = f"torch unavailable: {e}"
    if reason is None and not os.path.exists(os.path.join(ROOT, "gs2mesh_amd", "libgs2mesh_amd.so")):
        reason = "libgs2mesh_amd.so not built"
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Parity tests run twice: on the CPU emulator build of the kernel sources (not gpu) and on
    the real HIP library (-m gpu)."""
    from backends import make
    return make(request.param)
