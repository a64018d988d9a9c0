#!/usr/bin/env python3
"""Golden vectors for cpp_knn_batch_distance_pick, produced by the REFERENCE's own function
(knn_.cxx:138-203 compiled in place into oracle/_ref/libknn_ref.so) with the clock it seeds its
std::mt19937 from pinned by oracle/ref_shim.cpp.  Build container only (needs /root/reference).

    python tests/golden/make_golden_pick.py      -> tests/golden/knn_pick_small.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_harness as rh  # noqa: E402

out = {}
for name, (B, npts, nq, K, seed) in {"a": (2, 300, 150, 16, 20260925), "b": (1, 1000, 400, 8, 7),
                                      "c": (3, 64, 90, 4, 123456789)}.items():
    pts = np.random.RandomState(seed % 1000).rand(B, npts, 3).astype(np.float32)   # tie-free distances
    idx, q = rh.ref_knn_batch_distance_pick(pts, nq, K, seed)
    out.update({f"{name}/pts": pts, f"{name}/K": K, f"{name}/seed": seed, f"{name}/idx": idx, f"{name}/queries": q})
np.savez_compressed(os.path.join(HERE, "knn_pick_small.npz"), **out)
print("wrote knn_pick_small.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})
