"""ctypes binding of libffb6d_amd.so (the C ABI declared in include/ffb6d_knn.h and
include/ffb6d_ops.h).  There is NO fallback: if the library is missing or a call fails,
an exception is raised."""
import ctypes
import os

from . import build as _build

_c = ctypes
_vp, _i64, _i32, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t

# name -> (restype, argtypes); mirrors include/*.h one to one
SIGNATURES = {
    "ffb6d_last_error": (_c.c_char_p, []),
    "ffb6d_abi_version": (_c.c_int, []),
    "cpp_knn": (None, [_vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    "cpp_knn_omp": (None, [_vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    "cpp_knn_batch": (None, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "cpp_knn_batch_omp": (None, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "cpp_knn_batch_distance_pick": (None, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "cpp_knn_batch_distance_pick_omp": (None, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    "ffb6d_knn_workspace_bytes": (_sz, [_i64, _i64, _i64, _i32]),
    "ffb6d_knn_batch_device": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ffb6d_knn_prepared_bytes": (_sz, [_i64, _i64]),
    "ffb6d_knn_prepare_workspace_bytes": (_sz, [_i64, _i64]),
    "ffb6d_knn_prepare": (_i32, [_vp, _i64, _i64, _vp, _sz, _vp, _sz, _vp]),
    "ffb6d_knn_prepare_multi_workspace_bytes": (_sz, [_i32, _vp, _i64]),
    "ffb6d_knn_prepare_multi": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ffb6d_knn_search_multi": (_i32, [_i32, _vp, _i64, _vp]),
    "ffb6d_knn_search_prepared": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ffb6d_knn_set_pair_counter": (_i32, [_vp]),
    "ffb6d_knn_uses_pruning": (_i32, [_i64, _i64, _i64, _i32]),
    "ffb6d_random_sample_f32": (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_random_sample_bwd_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_nearest_interpolation_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_nearest_interpolation_bwd_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_gather_neighbour_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_gather_neighbour_bwd_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_relative_pos_encoding_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp]),
    "ffb6d_att_pool_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_relative_pos_encoding_cm_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp]),
    "ffb6d_att_pool_bwd_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_mlp_pm_f32": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _vp,
                                _i64, _i64, _i64, _i32, _i32, _vp]),
    "ffb6d_mlp_pm_tile": (_i32, [_i64, _i64, _i64, _i32]),
    "ffb6d_mlp_pm_choice": (_i32, [_i64, _i64, _i64, _i64, _i32, _i32, _i32]),
    "ffb6d_mlp_pm_seq_plan": (_i32, [_i64, _i64]),
    "ffb6d_mlp_pm_set_big_form": (None, [_i32]),
    "ffb6d_mlp_pm_set_seq_lin": (None, [_i32]),
    "ffb6d_att_pool_pm_f32": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_mlp_pm_bf16": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _vp,
                                 _i64, _i64, _i64, _i32, _i32, _vp]),
    "ffb6d_att_pool_pm_bf16": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_random_sample_pm": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_gather_rows_pm": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_relative_pos_encoding_pm": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp]),
    "ffb6d_affine_act_pm": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _c.c_float, _vp]),
    "ffb6d_affine_relu_maxpool_pm": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_bilinear_resize_pm": (_i32, [_i32, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_mlp_chain3_pm_f32": (_i32, [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _vp]),
    "ffb6d_mlp_chain3_pm_bf16": (_i32, [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _vp]),
    "ffb6d_upsampled_patch_rows_pm": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_posenc_mlp_pm": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _vp, _i32, _vp, _i64, _i64, _i32, _i64, _vp]),
    "ffb6d_lfa_pm": (_i32, [_i32, _i32, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64,
                            _i64, _i64, _i32, _i64, _i32, _vp]),
    "ffb6d_lfa_pm_choice": (_i32, [_i64, _i64, _i32]),
    "ffb6d_bilinear_bwd_pm": (_i32, [_i32, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_prelu_fwd": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp]),
    "ffb6d_prelu_bwd": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "ffb6d_gather_sum_rows": (_i32, [_i32, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp]),
    "ffb6d_random_sample_rows_bwd": (_i32, [_i32, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "ffb6d_att_pool_rows": (_i32, [_i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _vp]),
    "ffb6d_att_pool_rows_bwd": (_i32, [_i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i64, _vp]),
    "ffb6d_log_softmax_rows": (_i32, [_i32, _vp, _vp, _i64, _i64, _vp]),
    "ffb6d_log_softmax_rows_bwd": (_i32, [_i32, _vp, _vp, _vp, _i64, _i64, _vp]),
    "ffb6d_upconv_combine_pm": (_i32, [_i32, _vp, _vp, _c.c_float, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "ffb6d_upconv_set_form": (None, [_i32]),
    "ffb6d_pose_set_fit_form": (None, [_i32]),
    "ffb6d_pose_set_fit_spread": (None, [_i32]),
    "ffb6d_pose_set_big_form": (None, [_i32]),
    "ffb6d_psp_pool_pm_workspace_bytes": (_sz, [_i64, _i64, _i64, _vp, _i32]),
    "ffb6d_psp_pool_pm": (_i32, [_i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp, _sz, _vp]),
    "ffb6d_psp_prior_sum_pm": (_i32, [_i32, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i32, _vp]),
    "ffb6d_depth_to_cloud_f32": (_i32, [_vp, _vp, _c.c_float, _vp, _i64, _i64, _i64, _vp]),
    "ffb6d_pyramid_sets_f32": (_i32, [_vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "ffb6d_sample_points_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffb6d_sample_points_f32": (_i32, [_vp, _c.c_float, _vp, _vp, _i32, _vp, _c.c_uint64, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                       _i64, _vp, _sz, _vp]),
    "ffb6d_depth_normal": (_i32, [_vp, _i32, _c.c_double, _c.c_double, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _vp]),
    "ffb6d_fill_missing_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "ffb6d_fill_missing_f32": (_i32, [_vp, _c.c_double, _c.c_double, _c.c_float, _vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "ffb6d_check_index_range": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    # include/ffb6d_pose.h
    "ffb6d_vote_sets_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp, _vp, _vp]),
    "ffb6d_mean_shift_workspace_bytes": (_sz, [_i32, _i64]),
    "ffb6d_mean_shift_f32": (_i32, [_vp, _vp, _i32, _i32, _i64, _i64, _c.c_float, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                                    _sz, _vp]),
    "ffb6d_set_labels_to_points": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "ffb6d_refine_mask_by_center": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ffb6d_best_fit_transform_f32": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
}

_LIB = None


class FFB6DNativeError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load():
    """dlopen the library (once) and attach the prototypes.  Never builds implicitly:
    run `python -m ffb6d_amd.build` (or __graft_entry__.build()) first."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FFB6DNativeError(
            f"{path} is missing: build it with `python -m ffb6d_amd.build` "
            "(the HIP extension is mandatory, there is no CPU fallback)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def last_error():
    return load().ffb6d_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise FFB6DNativeError(f"{what} failed (rc={rc}): {last_error()}")


# ------------------------------------------------------------------------------------
# optional per-call HIP-event tracing (bench.py): when enabled for an op name, every native
# call of that op is bracketed by two events recorded on the stream the kernel is launched
# on, and tagged with its algorithmic byte count, so achieved GB/s can be computed live.
# ------------------------------------------------------------------------------------
class Tracer:
    def __init__(self, names=None, pred=None):
        self.names = set(names) if names else None   # None = every op
        self.pred = pred                              # optional (name, tag) -> bool: trace only these launches
        self.records = {}                             # name -> list of (start, end, bytes, tag)

    def wants(self, name, tag=None):
        return (self.names is None or name in self.names) and (self.pred is None or self.pred(name, tag))

    def summary(self, by_tag=False):
        """name -> dict(launches, total_ms, avg_us, bytes, gbps); call after a device sync.
        by_tag=True splits every op by its shape tag."""
        groups = {}
        for name, recs in self.records.items():
            for rec in recs:
                groups.setdefault((name, rec[3]) if by_tag else name, []).append(rec)
        out = {}
        for name, recs in groups.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            nbytes = sum(b for _, _, b, _ in recs)
            tot = sum(ms)
            out[name] = dict(launches=len(recs), total_ms=tot, avg_us=1e3 * tot / max(len(recs), 1),
                             bytes=nbytes, gbps=(nbytes / (tot * 1e-3) / 1e9) if tot > 0 else 0.0)
        return out


TRACER = None


class traced:
    """with traced("name", nbytes, tag): <native call>  -- no-op unless a Tracer is installed."""
    __slots__ = ("name", "nbytes", "tag", "start")

    def __init__(self, name, nbytes=0, tag=None):
        self.name, self.nbytes, self.tag, self.start = name, nbytes, tag, None

    def __enter__(self):
        t = TRACER
        if t is not None and t.wants(self.name, self.tag):
            import torch
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.start is not None:
            import torch
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            TRACER.records.setdefault(self.name, []).append((self.start, end, self.nbytes, self.tag))
        return False
