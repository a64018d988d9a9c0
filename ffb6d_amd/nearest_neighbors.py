"""Host-side mirror of the reference's Cython module `nearest_neighbors`
(ffb6d/models/RandLA/utils/nearest_neighbors/knn.pyx:32-109): same function names,
argument meaning, dtypes and output layout, backed by the gfx950 library.

    knn(pts [N,3], queries [Q,3], K, omp=False)         -> int64 [Q,K]
    knn_batch(pts [B,N,3], queries [B,Q,3], K, omp=False) -> int64 [B,Q,K]

numpy arrays go through the host-pointer C entry points `cpp_knn*` (same names and
signatures as knn_.h:4-19); `knn_batch_device` is the additional device-tensor entry the
reference lacks (torch tensors in/out, stream-ordered, no host round trip).
`knn_batch_distance_pick(pts, nqueries, K, omp=False)` mirrors knn.pyx:110-148 (unused by FFB6D; seeded
from time(0) like upstream unless FFB6D_KNN_PICK_SEED is set)."""
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


def knn_batch_distance_pick(pts, nqueries, K, omp=False):
    """knn.pyx:110-148: per frame draw `nqueries` query points (least-used first) and return
    (indices int64 [B,nqueries,K], queries float32 [B,nqueries,3])."""
    lib = _lib.load()
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    if pts_c.ndim != 3 or pts_c.shape[2] != 3:
        raise ValueError(f"expected [B,N,3] points, got {pts_c.shape}")
    K, nqueries = int(K), int(nqueries)
    if not 1 <= K <= 32 or pts_c.shape[1] < K:
        raise ValueError(f"need 1 <= K <= 32 and npts >= K (K={K}, npts={pts_c.shape[1]})")
    indices = np.full((pts_c.shape[0], nqueries, K), -1, dtype=np.int64)
    queries = np.zeros((pts_c.shape[0], nqueries, 3), dtype=np.float32)
    fn = lib.cpp_knn_batch_distance_pick_omp if omp else lib.cpp_knn_batch_distance_pick
    fn(pts_c.ctypes.data, pts_c.shape[0], pts_c.shape[1], 3, queries.ctypes.data, nqueries, K, indices.ctypes.data)
    if indices.size and indices.min() < 0:
        raise _lib.FFB6DNativeError("cpp_knn_batch_distance_pick failed: " + _lib.last_error())
    return indices, queries


class PreparedPoints:
    """A point set [B,S,3] in Morton order with per-tile bounding boxes (opaque device blob
    built by ffb6d_knn_prepare); reusable as support and/or query of knn_prepared()."""

    def __init__(self, points):
        import torch

        lib = _lib.load()
        if not points.is_cuda:
            raise _lib.FFB6DNativeError("PreparedPoints needs a GPU tensor (no CPU fallback)")
        if points.dtype != torch.float32 or points.dim() != 3 or points.shape[2] != 3:
            raise TypeError("points must be float32 [B,S,3]")
        self.points = points.contiguous()
        self.B, self.S = int(points.shape[0]), int(points.shape[1])
        if self.B < 1 or self.S < 1:
            raise ValueError("empty point set")
        nbytes = lib.ffb6d_knn_prepared_bytes(self.B, self.S)
        wbytes = lib.ffb6d_knn_prepare_workspace_bytes(self.B, self.S)
        self.blob = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
        ws = torch.empty((wbytes,), dtype=torch.uint8, device=points.device)
        with torch.cuda.device(points.device), _lib.traced("knn_prepare", 12 * self.B * self.S, (self.S,)):
            rc = lib.ffb6d_knn_prepare(self.points.data_ptr(), self.B, self.S, self.blob.data_ptr(), nbytes,
                                       ws.data_ptr(), wbytes, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "ffb6d_knn_prepare")


def prepare_many(point_sets):
    """Several point sets [B,S_i,3] (same B, same device) -> list of PreparedPoints, prepared TOGETHER by
    ffb6d_knn_prepare_multi: one Morton sort over the concatenation of all sets and one launch per remaining pass, instead
    of a dozen launches per set.  Byte-identical to PreparedPoints(p) for every p."""
    import ctypes

    import torch

    lib = _lib.load()
    sets = [p.contiguous() for p in point_sets]
    if not sets:
        return []
    B, dev = int(sets[0].shape[0]), sets[0].device
    for p in sets:
        if not p.is_cuda or p.dtype != torch.float32 or p.dim() != 3 or p.shape[2] != 3 or p.shape[0] != B or p.device != dev or p.shape[1] < 1:
            raise TypeError("prepare_many: float32 [B,S,3] GPU tensors of one batch size on one device")
    out = []
    for lo in range(0, len(sets), 8):                       # at most 8 sets per call
        chunk = sets[lo:lo + 8]
        n = len(chunk)
        S = (ctypes.c_int64 * n)(*[int(p.shape[1]) for p in chunk])
        nbytes = [lib.ffb6d_knn_prepared_bytes(B, int(p.shape[1])) for p in chunk]
        blobs = [torch.empty((nb,), dtype=torch.uint8, device=dev) for nb in nbytes]
        wbytes = lib.ffb6d_knn_prepare_multi_workspace_bytes(n, S, B)
        ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
        pts = (ctypes.c_void_p * n)(*[p.data_ptr() for p in chunk])
        prep = (ctypes.c_void_p * n)(*[b.data_ptr() for b in blobs])
        pbytes = (ctypes.c_size_t * n)(*nbytes)
        with torch.cuda.device(dev), _lib.traced("knn_prepare", 12 * B * sum(int(p.shape[1]) for p in chunk), tuple(int(p.shape[1]) for p in chunk)):
            rc = lib.ffb6d_knn_prepare_multi(n, pts, S, B, prep, pbytes, ws.data_ptr(), wbytes, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "ffb6d_knn_prepare_multi")
        for p, blob in zip(chunk, blobs):
            pp = PreparedPoints.__new__(PreparedPoints)
            pp.points, pp.B, pp.S, pp.blob = p, B, int(p.shape[1]), blob
            out.append(pp)
    return out


def knn_prepared(support, query, K, dtype=None, return_dist=False):
    """Exact KNN of `query` in a PreparedPoints `support` (same results as knn_batch_device).
    `query` is a PreparedPoints or -- for 2 <= K <= 16 -- a raw float32 [B,Q,3] GPU tensor."""
    import torch

    lib = _lib.load()
    raw = None
    if not isinstance(query, PreparedPoints):
        raw = query.contiguous()
        if raw.dtype != torch.float32 or raw.dim() != 3 or raw.shape[2] != 3 or not raw.is_cuda:
            raise TypeError("raw query must be a float32 [B,Q,3] GPU tensor")

        class _Q:      # shape carrier
            B, S = int(raw.shape[0]), int(raw.shape[1])
        query = _Q
        if not 2 <= int(K) <= 16:
            raise ValueError("raw queries need 2 <= K <= 16 (prepare the query set otherwise)")
    if support.B != query.B:
        raise ValueError("batch sizes differ")
    K = int(K)
    if not 1 <= K <= 32:
        raise ValueError(f"K must be in [1, 32], got {K}")
    if support.S < K:
        raise ValueError(f"npts ({support.S}) < K ({K})")
    dtype = dtype or torch.int64
    dev = support.points.device
    B, S, Q = support.B, support.S, query.S
    idx = torch.empty((B, Q, K), dtype=dtype, device=dev)
    dist = torch.empty((B, Q, K), dtype=torch.float32, device=dev) if return_dist else None
    nbytes = 12 * B * S + 12 * B * Q + idx.element_size() * B * Q * K
    with torch.cuda.device(dev), _lib.traced("knn", nbytes, (S, Q, K)):
        rc = lib.ffb6d_knn_search_prepared(
            support.blob.data_ptr(), None if raw is not None else query.blob.data_ptr(),
            raw.data_ptr() if raw is not None else None, B, S, Q, K,
            idx.data_ptr() if dtype == torch.int64 else None,
            idx.data_ptr() if dtype == torch.int32 else None,
            dist.data_ptr() if dist is not None else None, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "ffb6d_knn_search_prepared")
    return (idx, dist) if return_dist else idx


def search_many(searches, dtype=None):
    """Several independent searches over the same B frames in as few launches as kernels involved (ffb6d_knn_search_multi).
    `searches` = list of (support, query, K) with support / query a PreparedPoints or a raw float32 [B,S,3] GPU tensor; every
    search is routed exactly like the single calls (uses_pruning: Morton-ordered kernels on prepared sets -- K = 1 needs both
    sets prepared -- else the scan on the raw arrays).  Returns the index tensors [B,Q,K] in order."""
    import ctypes

    import torch

    lib = _lib.load()
    dtype = dtype or torch.int64
    if dtype not in (torch.int64, torch.int32):
        raise TypeError("index dtype must be int64 or int32")
    if not searches:
        return []

    class Search(ctypes.Structure):
        _fields_ = [("prep_support", ctypes.c_void_p), ("prep_query", ctypes.c_void_p), ("support", ctypes.c_void_p),
                    ("query", ctypes.c_void_p), ("S", ctypes.c_int64), ("Q", ctypes.c_int64), ("K", ctypes.c_int),
                    ("idx64", ctypes.c_void_p), ("idx32", ctypes.c_void_p), ("dist", ctypes.c_void_p)]

    def split(x):
        if isinstance(x, PreparedPoints):
            return x.blob.data_ptr(), x.points
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or x.shape[2] != 3:
            raise TypeError("point sets must be PreparedPoints or float32 [B,S,3] GPU tensors")
        return None, x.contiguous()

    arr = (Search * len(searches))()
    outs, keep, nbytes = [], [], 0
    B = None
    for i, (sup, qry, K) in enumerate(searches):
        ps, s_raw = split(sup)
        pq, q_raw = split(qry)
        K = _check(s_raw, q_raw, K, 3)
        B = int(s_raw.shape[0]) if B is None else B
        if s_raw.shape[0] != B:
            raise ValueError("search_many: every search must cover the same frames")
        S, Q = int(s_raw.shape[1]), int(q_raw.shape[1])
        idx = torch.empty((B, Q, K), dtype=dtype, device=s_raw.device)
        outs.append(idx)
        keep += [s_raw, q_raw]
        arr[i] = Search(ps, pq, s_raw.data_ptr(), q_raw.data_ptr(), S, Q, K, idx.data_ptr() if dtype == torch.int64 else None,
                        idx.data_ptr() if dtype == torch.int32 else None, None)
        nbytes += 12 * B * S + 12 * B * Q + idx.element_size() * B * Q * K       # 12 S + 12 Q + idx per search
    dev = outs[0].device
    with torch.cuda.device(dev), _lib.traced("knn", nbytes, tuple((int(a.S), int(a.Q), int(a.K)) for a in arr)):
        rc = lib.ffb6d_knn_search_multi(len(searches), arr, B, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "ffb6d_knn_search_multi")
    return outs


def uses_pruning(B, S, Q, K):
    return bool(_lib.load().ffb6d_knn_uses_pruning(B, S, Q, K))


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
    support_in = query if query is support else None   # self-KNN: one prepare serves both sides
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
    nbytes = 12 * B * S + 12 * B * Q + idx.element_size() * B * Q * K   # 12S + 12Q + idx
    with torch.cuda.device(support.device), _lib.traced("knn", nbytes, (S, Q, K)):
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.ffb6d_knn_batch_device(
            support.data_ptr(), (support if query is support_in else query).data_ptr(), B, S, Q, K,
            idx.data_ptr() if dtype == torch.int64 else None,
            idx.data_ptr() if dtype == torch.int32 else None,
            dist.data_ptr() if dist is not None else None,
            ws.data_ptr() if ws is not None else None, wbytes, stream)
    _lib.check(rc, "ffb6d_knn_batch_device")
    return (idx, dist) if return_dist else idx
