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
    """The package bound to the SIMT-emulated library (tests/simt: the product's kernel sources compiled for the host and run
    by a fiber-per-thread emulator), with the GPU guards of the host wrappers lifted so that CPU tensors reach the kernels:
    `_need_gpu` / `_stream` of the wrapper modules, `Tensor.is_cuda` (inline checks), and the handful of torch.cuda calls the
    wrappers make (device context, current stream).  Everything is undone when the module's tests are over."""
    import contextlib
    import ctypes

    import torch

    from ffb6d_amd import _lib, ops, ops_pm, pose
    from tests.simt import build
    lib = ctypes.CDLL(build.build())
    for name, (res, args) in _lib.SIGNATURES.items():        # every entry point the emulated sources export
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    mp = pytest.MonkeyPatch()
    mp.setattr(_lib, "_LIB", lib)
    for mod in (ops, ops_pm, pose):
        mp.setattr(mod, "_need_gpu", lambda *ts: None)
        mp.setattr(mod, "_stream", lambda t: None)
    mp.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())

    class FakeStream:                                        # kernels run synchronously in program order: streams and
        cuda_stream = None                                   # events only have to exist
        def __init__(self, device=None, priority=0, **kw):
            self.device = torch.device(device) if device is not None else torch.device("cpu")
        def wait_stream(self, other): pass
        def wait_event(self, event): pass
        def synchronize(self): pass
        def record_event(self, event=None): return event or FakeEvent()

    class FakeEvent:
        def __init__(self, enable_timing=False, **kw): pass
        def record(self, stream=None): pass
        def wait(self, stream=None): pass
        def synchronize(self): pass
        def query(self): return True
        def elapsed_time(self, other): return 0.0

    one_stream = FakeStream()
    mp.setattr(torch.cuda, "current_stream", lambda dev=None: one_stream)
    mp.setattr(torch.cuda, "Stream", FakeStream)
    mp.setattr(torch.cuda, "Event", FakeEvent)
    mp.setattr(torch.Tensor, "record_stream", lambda self, stream: None, raising=False)
    mp.setattr(torch.cuda, "current_device", lambda: 0)
    mp.setattr(torch.cuda, "stream", lambda st: contextlib.nullcontext())
    mp.setattr(torch.cuda, "synchronize", lambda dev=None: None)
    mp.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    request.addfinalizer(mp.undo)
    return lib
