import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _guard  # noqa: E402
_guard.maybe_install()   # FVK_GUARD_ALLOC=1 only: the guard-page device allocator, before the first device allocation (tests/_guard.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "reference: needs the read-only reference checkout at /root/reference")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are SKIPPED (not failed) on a box without a HIP device; on a GPU box nothing is skipped here, and the ops
    themselves fail loudly if libfvk_amd.so is missing."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
