import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--emulate", action="store_true", default=False,
                     help="run the -m gpu tests WITHOUT a GPU: CPU tensors, the package bound to the SIMT-emulated library "
                          "(tests/simt).  Functional pre-flight of the GPU suite; ~1e4 times slower than the device, so pick "
                          "tests with -k (full-size frames take hours)")


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


@pytest.fixture(scope="module")
def device(request):
    import torch
    if request.config.getoption("--emulate"):
        request.getfixturevalue("emu")          # binds the package to the emulated library for this module
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no ROCm device is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def emu(request):
    """The package bound to the SIMT-emulated library with the GPU guards of the host wrappers lifted so that CPU tensors reach
    the kernels (tests/simt/bind.py).  Everything is undone when the module's tests are over."""
    from tests.simt import bind
    mp = pytest.MonkeyPatch()
    lib = bind.bind(mp)
    request.addfinalizer(mp.undo)
    return lib
