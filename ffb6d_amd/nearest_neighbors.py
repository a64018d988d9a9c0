"""Host-side mirror of the reference's Cython module `nearest_neighbors`
(ffb6d/models/RandLA/utils/nearest_neighbors/knn.pyx:32-109): same function names,
argument meaning, dtypes and output layout, backed by the gfx950 library.

    knn(pts [N,3], queries [Q,3], K, omp=False)         -> int64 [Q,K]
    knn_batch(pts [B,N,3], queries [B,Q,3], K, omp=False) -> int64 [B,Q,K]

numpy arrays go through the host-pointer C entry points `cpp_knn*` (same names and
signatures as knn_.h:4-19); `knn_batch_device` is the additional device-tensor entry the
reference lacks (torch tensors in/out, stream-ordered, no host round trip).
`knn_batch_distance_pick` is deliberately absent (unused by FFB6D, time-seeded)."""
import numpy as np

from . import _lib


def _check(pts, queries, K, ndim):
    if pts.ndim != ndim or queries.ndim != ndim:
        raise ValueError(f"expected {ndim}-d arrays, got {pts.shape} and {queries.shape}")
    if pts.shape[-1] != 3 or queries.shape[-1] != 3:
        raise ValueError("ffb6d_amd KNN supports dim == 3 only")
    if ndim == 3 and pts.shape[0] != queries.shape[0]:
        raise ValueError("batch sizes differ")
    K = int(K)
    if not 1 <= K <= 32:
        raise ValueError(f"K must be in [1, 32], got {K}")
    if pts.shape[-2] < K:
        raise ValueError(f"npts ({pts.shape[-2]}) < K ({K}): undefined in the reference (knn_.cxx:121-131)")
    return K


def knn(pts, queries, K, omp=False):
    lib = _lib.load()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = np.ascontiguousarray(queries, dtype=np.float32)
    K = _check(pts_c, q_c, K, 2)
    indices = np.full((q_c.shape[0], K), -1, dtype=np.int64)
    fn = lib.cpp_knn_omp if omp else lib.cpp_knn
    fn(pts_c.ctypes.data, pts_c.shape[0], 3, q_c.ctypes.data, q_c.shape[0], K, indices.ctypes.data)
    if indices.size and indices.min() < 0:   # the C signature is void: detect untouched output
        raise _lib.FFB6DNativeError("cpp_knn failed: " + _lib.last_error())
    return indices


def knn_batch(pts, queries, K, omp=False):
    lib = _lib.load()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = np.ascontiguousarray(queries, dtype=np.float32)
    K = _check(pts_c, q_c, K, 3)
    indices = np.full((pts_c.shape[0], q_c.shape[1], K), -1, dtype=np.int64)
    fn = lib.cpp_knn_batch_omp if omp else lib.cpp_knn_batch
    fn(pts_c.ctypes.data, pts_c.shape[0], pts_c.shape[1], 3, q_c.ctypes.data, q_c.shape[1], K,
       indices.ctypes.data)
    if indices.size and indices.min() < 0:
        raise _lib.FFB6DNativeError("cpp_knn_batch failed: " + _lib.last_error())
    return indices


def knn_batch_device(support, query, K, dtype=None, return_dist=False):
    """Device entry: support [B,S,3], query [B,Q,3] float32 CUDA(HIP) tensors ->
    index tensor [B,Q,K] (torch.int64 default, or torch.int32) on the same device,
    optionally also the squared distances [B,Q,K] f32."""
    import torch

    lib = _lib.load()
    if not (support.is_cuda and query.is_cuda):
        raise _lib.FFB6DNativeError("knn_batch_device needs GPU tensors (no CPU fallback)")
    if support.dtype != torch.float32 or query.dtype != torch.float32:
        raise TypeError("support/query must be float32")
    support = support.contiguous()
    query = query.contiguous()
    K = _check(support, query, K, 3)
    dtype = dtype or torch.int64
    if dtype not in (torch.int64, torch.int32):
        raise TypeError("index dtype must be int64 or int32")
    B, S, _ = support.shape
    Q = query.shape[1]
    idx = torch.empty((B, Q, K), dtype=dtype, device=support.device)
    dist = torch.empty((B, Q, K), dtype=torch.float32, device=support.device) if return_dist else None
    wbytes = lib.ffb6d_knn_workspace_bytes(B, S, Q, K)
    ws = torch.empty((wbytes,), dtype=torch.uint8, device=support.device) if wbytes else None
    with torch.cuda.device(support.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.ffb6d_knn_batch_device(
            support.data_ptr(), query.data_ptr(), B, S, Q, K,
            idx.data_ptr() if dtype == torch.int64 else None,
            idx.data_ptr() if dtype == torch.int32 else None,
            dist.data_ptr() if dist is not None else None,
            ws.data_ptr() if ws is not None else None, wbytes, stream)
    _lib.check(rc, "ffb6d_knn_batch_device")
    return (idx, dist) if return_dist else idx
