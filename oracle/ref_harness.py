"""oracle/ref_harness.py -- TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).

Imports the *reference's own* Python hot path read-only from /root/reference so that
  - our restatements (oracle/*.py, oracle/knn_oracle.c) can be pinned against it, and
  - golden vectors can be generated (tests/golden/make_golden.py).

Nothing is copied out of the reference tree.  Three harness-side shims make the import
possible on this image:
  1. `cpp_wrappers.cpp_subsampling.grid_subsampling` (helper_tool.py:14) does not compile
     against numpy 2.x and is never called by FFB6D -> empty stub modules;
  2. `nearest_neighbors.lib.python.nearest_neighbors` (helper_tool.py:15) -> a stub module
     whose knn/knn_batch call the reference's own C++ (knn_.cxx compiled in place into
     oracle/_ref/libknn_ref.so by oracle/Makefile) with the marshalling of knn.pyx:32-109;
  3. `torch.utils.model_zoo.load_url` (extractors.py:216-222 downloads ResNet34 weights)
     -> returns the randomly initialised state dict (weights are overwritten by the
     deterministic synthetic weights of ffb6d_amd.synth anyway).
"""
import ctypes
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("FFB6D_REFERENCE", "/root/reference")
REF_FFB6D = os.path.join(REF_ROOT, "ffb6d")
HERE = os.path.dirname(os.path.abspath(__file__))
REF_KNN_SO = os.path.join(HERE, "_ref", "libknn_ref.so")


def reference_available():
    return os.path.isdir(os.path.join(REF_FFB6D, "models"))


_ref_knn_lib = None


def ref_knn_lib():
    """The reference's own compiled KNN (oracle/_ref/libknn_ref.so)."""
    global _ref_knn_lib
    if _ref_knn_lib is None:
        if not os.path.exists(REF_KNN_SO):
            raise RuntimeError(f"{REF_KNN_SO} missing: run `make -C oracle` where /root/reference exists")
        lib = ctypes.CDLL(REF_KNN_SO)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        for name in ("ref_cpp_knn", "ref_cpp_knn_omp"):
            getattr(lib, name).argtypes = [vp, sz, sz, vp, sz, sz, vp]
            getattr(lib, name).restype = None
        for name in ("ref_cpp_knn_batch", "ref_cpp_knn_batch_omp"):
            getattr(lib, name).argtypes = [vp, sz, sz, sz, vp, sz, sz, vp]
            getattr(lib, name).restype = None
        lib.ref_cpp_knn_batch_distance_pick.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp]
        lib.ref_cpp_knn_batch_distance_pick.restype = None
        lib.ref_set_fixed_time.argtypes = [ctypes.c_long]
        lib.ref_set_fixed_time.restype = None
        _ref_knn_lib = lib
    return _ref_knn_lib


def ref_knn_batch_distance_pick(pts, nqueries, K, seed):
    """knn.pyx:110-148 marshalling around the reference's cpp_knn_batch_distance_pick, with the clock the
    reference seeds its std::mt19937 from (time(0), knn_.cxx:143) pinned to `seed` by oracle/ref_shim.cpp."""
    lib = ref_knn_lib()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    B, npts, dim = pts_c.shape
    idx = np.zeros((B, nqueries, K), dtype=np.int64)
    queries = np.zeros((B, nqueries, dim), dtype=np.float32)
    lib.ref_set_fixed_time(int(seed))
    try:
        lib.ref_cpp_knn_batch_distance_pick(pts_c.ctypes.data, B, npts, dim, queries.ctypes.data, nqueries, K,
                                            idx.ctypes.data)
    finally:
        lib.ref_set_fixed_time(-1)
    return idx, queries


def ref_knn_batch(pts, queries, K, omp=False):
    """knn.pyx:71-109 marshalling around the reference's cpp_knn_batch[_omp]."""
    lib = ref_knn_lib()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = np.ascontiguousarray(queries, dtype=np.float32)
    B, npts, dim = pts_c.shape
    nq = q_c.shape[1]
    out = np.zeros((B, nq, K), dtype=np.int64)
    fn = lib.ref_cpp_knn_batch_omp if omp else lib.ref_cpp_knn_batch
    fn(pts_c.ctypes.data, B, npts, dim, q_c.ctypes.data, nq, K, out.ctypes.data)
    return out


def ref_knn(pts, queries, K, omp=False):
    """knn.pyx:32-69 marshalling around the reference's cpp_knn[_omp]."""
    lib = ref_knn_lib()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = np.ascontiguousarray(queries, dtype=np.float32)
    npts, dim = pts_c.shape
    nq = q_c.shape[0]
    out = np.zeros((nq, K), dtype=np.int64)
    fn = lib.ref_cpp_knn_omp if omp else lib.ref_cpp_knn
    fn(pts_c.ctypes.data, npts, dim, q_c.ctypes.data, nq, K, out.ctypes.data)
    return out


_installed = False


def install():
    """Make `import models.ffb6d` / `import helper_tool` resolve to the reference tree."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_FFB6D}")
    import torch
    import torch.utils.model_zoo as model_zoo

    # shim 1: never-called grid subsampling extension
    for name in ("cpp_wrappers", "cpp_wrappers.cpp_subsampling",
                 "cpp_wrappers.cpp_subsampling.grid_subsampling"):
        sys.modules.setdefault(name, types.ModuleType(name))
    # shim 2: the Cython KNN module, backed by the reference's own C++
    nn_mod = types.ModuleType("nearest_neighbors.lib.python.nearest_neighbors")
    nn_mod.knn = ref_knn
    nn_mod.knn_batch = ref_knn_batch
    for name in ("nearest_neighbors", "nearest_neighbors.lib", "nearest_neighbors.lib.python"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nearest_neighbors.lib.python.nearest_neighbors"] = nn_mod
    sys.modules["nearest_neighbors.lib.python"].nearest_neighbors = nn_mod
    sys.modules["nearest_neighbors.lib"].python = sys.modules["nearest_neighbors.lib.python"]
    sys.modules["nearest_neighbors"].lib = sys.modules["nearest_neighbors.lib"]
    # shim 3: no network
    model_zoo.load_url = lambda *a, **k: _Resnet34Init.state
    sys.path.insert(0, REF_FFB6D)
    sys.path.insert(0, os.path.join(REF_FFB6D, "models", "RandLA"))
    _installed = True


class _Resnet34Init:
    state = None


def build_reference_model(n_classes=22, n_pts=12288, n_kps=8):
    """The unmodified reference FFB6D (ffb6d/models/ffb6d.py:16-19), eval mode, CPU."""
    install()
    cwd = os.getcwd()
    os.chdir(REF_FFB6D)  # common.py:165 resolves relative paths at import
    try:
        import models.cnn.extractors as extractors
        if _Resnet34Init.state is None:
            _Resnet34Init.state = extractors.ResNet(extractors.BasicBlock, [3, 4, 6, 3]).state_dict()
        from common import ConfigRandLA
        from models.ffb6d import FFB6D
        rndla_cfg = ConfigRandLA
        model = FFB6D(n_classes=n_classes, n_pts=n_pts, rndla_cfg=rndla_cfg, n_kps=n_kps)
    finally:
        os.chdir(cwd)
    return model.eval()


def reference_modules():
    """(models.ffb6d, models.RandLA.RandLANet, helper_tool) of the reference."""
    install()
    cwd = os.getcwd()
    os.chdir(REF_FFB6D)
    try:
        import models.ffb6d as m_ffb6d
        import models.RandLA.RandLANet as m_randla
        import helper_tool
    finally:
        os.chdir(cwd)
    return m_ffb6d, m_randla, helper_tool


def reference_pose_modules(mesh_kps=None, mesh_ctr=None, r_lst=None):
    """(utils.meanshift_pytorch, utils.pvn3d_eval_utils_kpls) of the reference, importable on this
    CPU-only image through harness-side shims (nothing in the reference is edited):
      4. `cv2`, `plyfile`, `normalSpeed` (meanshift_pytorch.py:3, basic_utils.py:8-9: display, mesh
         files and normal estimation -- none is on the pose-solver path) -> empty stub modules;
      5. `torch.Tensor.cuda` -> identity, so `torch.zeros(...).cuda()` (pvn3d_eval_utils_kpls.py:79,235)
         stays on the CPU;
      6. the module-level `bs_utils` / `bs_utils_lm` / `config.ycb_r_lst` (dataset-file readers,
         pvn3d_eval_utils_kpls.py:18-25) -> objects serving the synthetic mesh keypoints handed in:
         mesh_kps [n_cls,n_kps,3], mesh_ctr [n_cls,3] indexed by class id; r_lst[cls_id-1]."""
    install()
    import torch
    for name in ("cv2", "plyfile", "normalSpeed"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.imshow = mod.waitKey = lambda *a, **k: None
            mod.PlyData = object
            sys.modules[name] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    os.chdir(REF_FFB6D)
    try:
        import utils.meanshift_pytorch as ms_mod
        import utils.pvn3d_eval_utils_kpls as pose_mod
    finally:
        os.chdir(cwd)

    class _Mesh:
        def __init__(self, by_name):
            self.by_name = by_name

        def _cls(self, key):
            return pose_mod.cls_lst.index(key) + 1 if self.by_name else int(key)

        def get_kps(self, key, **kw):
            return np.asarray(mesh_kps[self._cls(key)], np.float32).copy()

        def get_ctr(self, key, **kw):
            return np.asarray(mesh_ctr[self._cls(key)], np.float32).copy()

    if mesh_kps is not None:
        pose_mod.bs_utils = _Mesh(by_name=True)          # YCB path looks objects up by name (:149-152)
        pose_mod.bs_utils_lm = _Mesh(by_name=False)      # LineMOD path by obj_id (:277-280)
    if r_lst is not None:
        pose_mod.config.ycb_r_lst = list(np.asarray(r_lst, np.float64))
    return ms_mod, pose_mod


def reference_dataset_class():
    """The reference's LineMOD `Dataset` class (ffb6d/datasets/linemod/linemod_dataset.py:25) importable without its
    third-party stack: `cv2`, `torchvision(.transforms)`, `termcolor`, `normalSpeed`, `plyfile` -> stub modules (none of
    them is touched by the methods the tests call UNBOUND: `dpt_2_pcld`, :188-199).  Nothing in the reference is edited."""
    install()
    for name in ("cv2", "torchvision", "torchvision.transforms", "termcolor", "normalSpeed", "plyfile"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.imshow = mod.waitKey = mod.colored = lambda *a, **k: None
            mod.PlyData = object
            sys.modules[name] = mod
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # by file path: an installed `datasets` distribution (HuggingFace) shadows the reference's namespace package
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ffb6d_reference_linemod_dataset", os.path.join(REF_FFB6D, "datasets", "linemod", "linemod_dataset.py"))
    lm = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    os.chdir(REF_FFB6D)
    try:
        spec.loader.exec_module(lm)
    finally:
        os.chdir(cwd)
    return lm.Dataset
