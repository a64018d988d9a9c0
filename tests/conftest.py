import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import ref_harness
    if not ref_harness.reference_available():
        skip = pytest.mark.skip(reason="/root/reference not present on this machine")
        for item in items:
            if "reference" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def native_lib():
    """The built HIP library; builds it in-tree when missing (hipcc cross-compiles on CPU)."""
    from ffb6d_amd import _lib, build
    if build.needs_build():
        build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no ROCm device is visible")
    return torch.device("cuda:0")
