"""Bind the package to the SIMT-emulated library (tests/simt: the product's kernel sources compiled for the host and run by a
fiber-per-thread emulator) and lift the GPU guards of the host wrappers so that CPU tensors reach the kernels: `_need_gpu` /
`_stream` of the wrapper modules, `Tensor.is_cuda` (inline checks), and the handful of torch.cuda calls the wrappers make (device
context, current stream).  Test infrastructure: used by the `emu` fixture (tests/conftest.py) and by worker processes of the
multi-process CPU tests."""
import contextlib
import ctypes


def bind(mp):
    """mp: a pytest.MonkeyPatch (undo() restores everything).  Returns the loaded emulator library."""
    import torch

    from ffb6d_amd import _lib, ops, ops_cl, ops_pm, pose
    from tests.simt import build
    lib = ctypes.CDLL(build.build())
    for name, (res, args) in _lib.SIGNATURES.items():        # every entry point the emulated sources export
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    mp.setattr(_lib, "_LIB", lib)
    for mod in (ops, ops_cl, ops_pm, pose):
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
    return lib
