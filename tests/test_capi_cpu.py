"""CPU suite: the C-ABI library builds, loads and exports every symbol that include/*.h
declares (no compute calls -- there is no GPU here), and the Python host layer fails
loudly instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def declared_functions():
    names = []
    for hdr in sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\(", src, flags=re.M):
            name = m.group(1)
            if name.startswith(("ffb6d_", "cpp_knn")):
                names.append(name)
    return sorted(set(names))


def test_headers_declare_the_expected_entry_points():
    names = declared_functions()
    for must in ("cpp_knn", "cpp_knn_omp", "cpp_knn_batch", "cpp_knn_batch_omp",
                 "cpp_knn_batch_distance_pick", "cpp_knn_batch_distance_pick_omp",
                 "ffb6d_knn_batch_device", "ffb6d_random_sample_f32",
                 "ffb6d_nearest_interpolation_f32", "ffb6d_gather_neighbour_f32",
                 "ffb6d_relative_pos_encoding_f32", "ffb6d_att_pool_f32", "ffb6d_mlp_pm_f32",
                 "ffb6d_vote_sets_f32", "ffb6d_mean_shift_f32", "ffb6d_mean_shift_workspace_bytes",
                 "ffb6d_set_labels_to_points", "ffb6d_refine_mask_by_center", "ffb6d_best_fit_transform_f32"):
        assert must in names


def test_library_exports_every_declared_symbol(native_lib):
    from ffb6d_amd import _lib
    for name in declared_functions():
        assert hasattr(native_lib, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert native_lib.ffb6d_abi_version() >= 1000


@pytest.mark.reference
def test_reference_cython_binding_builds_and_links_against_this_library(native_lib, tmp_path):
    """The reference's own knn.pyx (knn.pyx:7-148: `cdef extern from "knn_.h"`, all six cpp_knn* symbols), taken
    verbatim except for its stale line 2 (`# distutils: sources = knn.cxx` names a file that does not exist
    upstream either), compiles against include/knn_.h and links against libffb6d_amd.so -- no knn_.cxx, no
    nanoflann.  Import succeeds, i.e. every symbol the binding declares resolves in this library."""
    import subprocess
    import sys
    import sysconfig
    from ffb6d_amd import build
    ref = "/root/reference/ffb6d/models/RandLA/utils/nearest_neighbors/knn.pyx"
    lines = open(ref).read().split("\n")
    assert lines[1].startswith("# distutils: sources")
    (tmp_path / "nearest_neighbors.pyx").write_text("\n".join(lines[:1] + lines[2:]))
    subprocess.run([sys.executable, "-m", "cython", "--cplus", "-3", "nearest_neighbors.pyx"], cwd=tmp_path, check=True)
    so = tmp_path / ("nearest_neighbors" + sysconfig.get_config_var("EXT_SUFFIX"))
    lib_dir = os.path.dirname(build.LIB_PATH)
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-w", "-std=c++11", "nearest_neighbors.cpp", "-o", str(so),
                    "-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"],
                    "-I" + np.get_include(), "-L" + lib_dir, "-lffb6d_amd", "-Wl,-rpath," + lib_dir,
                    "-Wl,--no-undefined", "-L" + sysconfig.get_config_var("LIBDIR"),
                    "-lpython" + sysconfig.get_config_var("LDVERSION")], cwd=tmp_path, check=True)
    code = ("import sys; sys.path.insert(0, %r); import nearest_neighbors as m; "
            "print(all(hasattr(m, n) for n in ('knn', 'knn_batch', 'knn_batch_distance_pick')))" % str(tmp_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert res.stdout.strip() == "True", res.stdout + res.stderr


def test_pose_entry_points_validate_arguments_without_a_gpu(native_lib):
    """Argument checks run before any HIP call: bad sizes come back as FFB6D_ERR_ARG with a message."""
    from ffb6d_amd import _lib
    assert native_lib.ffb6d_mean_shift_workspace_bytes(0, 128) == 0
    need = native_lib.ffb6d_mean_shift_workspace_bytes(9, 1024)
    assert need >= 2 * 9 * 1024 * 16
    rc = native_lib.ffb6d_mean_shift_f32(None, None, 1, 4, 64, 0, -1.0, 300, 8, None, None, None, None, None, 0, None)
    assert rc == -1 and "bandwidth" in _lib.last_error()
    rc = native_lib.ffb6d_vote_sets_f32(None, None, None, 16, None, None, None, 1, 1, 1, 8, 8, None, None, None)
    assert rc == -1 and "mask_bits" in _lib.last_error()
    rc = native_lib.ffb6d_vote_sets_f32(None, None, None, 64, None, None, None, 1, 1, 1, 8, 4, None, None, None)
    assert rc == -1 and "set_stride" in _lib.last_error()
    assert native_lib.ffb6d_best_fit_transform_f32(None, None, 0, 9, None, None) == 0      # nothing to do
    assert native_lib.ffb6d_best_fit_transform_f32(None, None, 2, 0, None, None) == -1


def test_pyramid_sets_validates_arguments_without_a_gpu(native_lib):
    import ctypes
    from ffb6d_amd import _lib
    f = native_lib.ffb6d_pyramid_sets_f32
    assert f(None, 0, 0, 0, 0, 0, 0, None, None, None, None, 0, 0, 0, None, None, None) == 0          # nothing to do
    assert f(None, 0, 0, 0, 2, 16, 7, None, None, None, None, 0, 0, 0, None, None, None) == -1 and "at most" in _lib.last_error()
    n = (ctypes.c_int64 * 1)(16)
    out = (ctypes.c_void_p * 1)(None)
    assert f(None, 48, 1, 3, 2, 16, 1, n, out, None, None, 0, 0, 0, None, None, None) == -1 and "null cloud" in _lib.last_error()
    st = (ctypes.c_int * 2)(2, 3)
    g = (ctypes.c_void_p * 2)(64, 64)
    assert f(None, 0, 0, 0, 1, 0, 0, None, None, None, ctypes.c_void_p(64), 8, 8, 2, st, g, None) == -1 and "multiple" in _lib.last_error()


def test_workspace_query_is_pure_host_logic(native_lib):
    # small support, many query blocks -> brute-force scan without split, no scratch
    assert native_lib.ffb6d_knn_uses_pruning(64, 256, 12288, 16) == 0
    assert native_lib.ffb6d_knn_workspace_bytes(64, 256, 12288, 16) == 0
    # from 512 support points on the Morton-prepared search is the faster one
    assert native_lib.ffb6d_knn_uses_pruning(64, 1024, 12288, 16) == 1
    # big support -> Morton-prepared sets + sort scratch
    assert native_lib.ffb6d_knn_uses_pruning(8, 76800, 768, 16) == 1
    need = native_lib.ffb6d_knn_workspace_bytes(8, 76800, 768, 16)
    assert need >= native_lib.ffb6d_knn_prepared_bytes(8, 76800) + native_lib.ffb6d_knn_prepared_bytes(8, 768)
    assert native_lib.ffb6d_knn_workspace_bytes(0, 10, 10, 16) == 0


def test_pose_solver_workspace_and_form_switches_are_host_logic(native_lib):
    """Round 6: the mean-shift workspace also holds the representative / owner maps (u32 per point each) and five ints per set of the
    chip-wide rounds of large vote sets; the form switches are plain setters (pose.set_fit_spread keeps a Python-side mirror that
    pipeline.SensorToPose.run uses to switch the chip-wide first rounds off under the overlapped schedule and restore them)."""
    from ffb6d_amd import pose
    a256 = lambda n: (n + 255) // 256 * 256                                      # noqa: E731
    G, stride = 40, 12288
    want = 2 * a256(G * stride * 16) + a256(3 * 4 * G) + a256(4 * G) + a256(8 * G) + 2 * a256(G * stride * 4) + a256(5 * 4 * G)
    assert native_lib.ffb6d_mean_shift_workspace_bytes(G, stride) == want
    assert native_lib.ffb6d_mean_shift_workspace_bytes(0, stride) == 0
    assert pose.FIT_SPREAD == 1
    assert pose.set_fit_spread(0) == 1 and pose.FIT_SPREAD == 0
    assert pose.set_fit_spread(1) == 0 and pose.FIT_SPREAD == 1
    for setter in ("ffb6d_pose_set_fit_form", "ffb6d_pose_set_big_form"):
        getattr(native_lib, setter)(0)
        getattr(native_lib, setter)(1)


def test_argument_errors_are_reported_without_a_gpu(native_lib):
    from ffb6d_amd import _lib
    rc = native_lib.ffb6d_knn_batch_device(None, None, 1, 100, 10, 64, None, None, None, None, 0, None)
    assert rc != 0 and "K must be" in _lib.last_error()
    rc = native_lib.ffb6d_knn_batch_device(None, None, 1, 8, 10, 16, None, None, None, None, 0, None)
    assert rc != 0 and "npts" in _lib.last_error()
    rc = native_lib.ffb6d_random_sample_f32(None, None, 16, None, None, 1, 1, 1, 1, 16, None)
    assert rc != 0 and "idx_bits" in _lib.last_error()
    # round 4's entry points: shapes are checked before anything touches a device
    rc = native_lib.ffb6d_mlp_chain3_pm_f32(None, 128, None, None, 1, None, None, 1, None, None, 0, None, 24, 100, 22, None)
    assert rc != 0 and "cout3" in _lib.last_error()                                  # 22 is not a multiple of 4
    rc = native_lib.ffb6d_mlp_chain3_pm_f32(None, 128, None, None, 1, None, None, 3, None, None, 0, None, 24, 100, 24, None)
    assert rc != 0 and "act" in _lib.last_error()                                    # no log-softmax inside the chain
    assert native_lib.ffb6d_mlp_chain3_pm_f32(None, 128, None, None, 1, None, None, 1, None, None, 0, None, 24, 0, 24, None) == 0      # no rows
    rc = native_lib.ffb6d_upsampled_patch_rows_pm(0, None, None, 16, None, 1, 4, 4, 8, 8, 8, 10, None)
    assert rc != 0 and "idx_bits" in _lib.last_error()
    rc = native_lib.ffb6d_upsampled_patch_rows_pm(0, None, None, 64, None, 1, 4, 4, 8, 8, 6, 10, None)
    assert rc != 0 and "bad shape" in _lib.last_error()                              # 6 channels: not whole 16-byte units
    assert native_lib.ffb6d_upsampled_patch_rows_pm(0, None, None, 64, None, 1, 4, 4, 8, 8, 8, 0, None) == 0                           # no rows


def test_ops_refuse_cpu_tensors():
    from ffb6d_amd import _lib, ops
    f = torch.zeros(1, 4, 10, 1)
    i = torch.zeros(1, 3, 16, dtype=torch.int64)
    with pytest.raises(_lib.FFB6DNativeError):
        ops.random_sample(f, i)
    with pytest.raises(_lib.FFB6DNativeError):
        ops.nearest_interpolation(f, torch.zeros(1, 5, 1, dtype=torch.int64))
    with pytest.raises(_lib.FFB6DNativeError):
        ops.gather_neighbour(torch.zeros(1, 10, 4), i)
    with pytest.raises(_lib.FFB6DNativeError):
        ops.att_pool(torch.zeros(1, 2, 3, 16), torch.zeros(1, 2, 3, 16))


def test_knn_python_boundary_validates_like_the_header_says(native_lib):
    from ffb6d_amd import nearest_neighbors as nn
    pts = np.zeros((2, 10, 3), np.float32)
    with pytest.raises(ValueError):
        nn.knn_batch(pts, pts, 16)           # npts < K
    with pytest.raises(ValueError):
        nn.knn_batch(pts, pts[:1], 4)        # batch mismatch
    with pytest.raises(ValueError):
        nn.knn_batch(np.zeros((2, 10, 4), np.float32), np.zeros((2, 10, 4), np.float32), 4)  # dim != 3
    with pytest.raises(ValueError):
        nn.knn(np.zeros((10, 3), np.float32), np.zeros((10, 3), np.float32), 33)


def test_synthetic_frames_are_deterministic():
    from ffb6d_amd import synth
    a = synth.make_frame(2000, n_points=768, height=120, width=160)
    b = synth.make_frame(2000, n_points=768, height=120, width=160)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["cld"].shape == (768, 3) and a["cld_rgb_nrm"].shape == (9, 768)
    assert (a["cld"][:, 2] > 0).all()  # only valid-depth pixels are chosen
    g = synth.strided_grids(a["dpt_xyz"])
    assert g[4].shape == (30 * 40, 3) and g[8].shape == (15 * 20, 3)
    np.testing.assert_array_equal(g[4][41], a["dpt_xyz"][:, 4, 4])  # pixel (y*s, x*s), row-major
