"""oracle/knn.py -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/knn_oracle.c (our
CPU restatement of knn_.cxx:104-135) with the marshalling of knn.pyx:71-109 and of
DataProcessing.knn_search (helper_tool.py:160-170)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle_knn.so")
_lib = None


def build():
    """(Re)build the C restatement and, where /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        l = ctypes.CDLL(SO)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        l.oracle_knn_batch.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp]
        l.oracle_knn_batch.restype = None
        l.oracle_knn_batch_dist.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp, vp]
        l.oracle_knn_batch_dist.restype = None
        l.oracle_knn_batch_distance_pick.argtypes = [vp, sz, sz, sz, vp, sz, sz, vp, ctypes.c_uint32]
        l.oracle_knn_batch_distance_pick.restype = None
        _lib = l
    return _lib


def knn_batch(pts, queries, K, omp=False, return_dist=False):
    """knn.pyx:71-109: float32 contiguous copies in, zero-initialised int64 [B,Q,K] out."""
    p = np.ascontiguousarray(pts, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    B, npts, dim = p.shape
    nq = q.shape[1]
    idx = np.zeros((B, nq, K), dtype=np.int64)
    if return_dist:
        dist = np.zeros((B, nq, K), dtype=np.float32)
        lib().oracle_knn_batch_dist(p.ctypes.data, B, npts, dim, q.ctypes.data, nq, K,
                                    idx.ctypes.data, dist.ctypes.data)
        return idx, dist
    lib().oracle_knn_batch(p.ctypes.data, B, npts, dim, q.ctypes.data, nq, K, idx.ctypes.data)
    return idx


def knn_batch_distance_pick(pts, nqueries, K, seed, omp=False):
    """knn.pyx:110-148 around the restatement of cpp_knn_batch_distance_pick (knn_.cxx:138-203), with the
    std::mt19937 seed passed in instead of time(0).  Returns (indices int64 [B,nq,K], queries f32 [B,nq,3])."""
    p = np.ascontiguousarray(pts, dtype=np.float32)
    B, npts, dim = p.shape
    idx = np.zeros((B, nqueries, K), dtype=np.int64)
    queries = np.zeros((B, nqueries, dim), dtype=np.float32)
    lib().oracle_knn_batch_distance_pick(p.ctypes.data, B, npts, dim, queries.ctypes.data, nqueries, K,
                                         idx.ctypes.data, int(seed) & 0xffffffff)
    return idx, queries


def knn(pts, queries, K, omp=False):
    """knn.pyx:32-69 (single cloud)."""
    return knn_batch(np.asarray(pts)[None], np.asarray(queries)[None], K)[0]


def knn_search(support_pts, query_pts, k):
    """helper_tool.py:160-170."""
    return knn_batch(support_pts, query_pts, k, omp=True).astype(np.int32)


def sqdist_f32(q, p):
    """((dx*dx + dy*dy) + dz*dz) with every op rounded to f32 (nanoflann.hpp:323-348)."""
    q = np.asarray(q, np.float32)
    p = np.asarray(p, np.float32)
    d = q - p
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def canonical_ties(idx, support, query):
    """Re-order every run of EQUAL distances inside a KNN row by ascending index.

    The reference's kd-tree keeps the first-VISITED candidate of an exact distance tie
    (nanoflann.hpp:118-135), which depends on the tree traversal; the oracle and the HIP
    kernel keep the lowest index.  Both orders list the same neighbours at the same
    distances; this maps either of them to one canonical form so they can be compared
    bit-exactly.  idx [Q,K] into support [S,3]; query [Q,3].  Returns (canonical idx,
    number of rows that contained a tie)."""
    idx = np.asarray(idx)
    d = sqdist_f32(np.asarray(query)[:, None, :], np.asarray(support)[idx.astype(np.int64)])
    order = np.lexsort((idx, d), axis=-1)
    canon = np.take_along_axis(idx, order, axis=-1)
    ties = int(((np.diff(d, axis=-1) == 0).any(axis=-1)).sum()) if idx.shape[-1] > 1 else 0
    return canon, ties
