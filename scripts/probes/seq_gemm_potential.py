#!/usr/bin/env python3
"""What hiding the epilogue is worth with the tail quantisation taken out: rows = 128 * 512 (one point tile per slot of the chip), so the
LDS-tiled form runs whole rounds of 512 workgroups and the tile-sequence form exactly one round.  python scripts/seq_gemm_potential.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import ops_pm
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = 128 * 512
for K in (128, 256, 1024):
    for C, ts in ((256, (2,)), (512, (2, 4)), (1024, (2, 4, 8))):
        x = torch.randn(R, K, device=dev); w = torch.randn(C, K, device=dev) / K ** 0.5; b = torch.randn(C, device=dev)
        out = torch.empty(R, C, device=dev)
        vs = [7] + [8 + 256 * t for t in ts]
        tm = {v: [] for v in vs}
        for v in vs:
            ops_pm.mlp(x, w, b, 1, out=out, tile_hint=v)
        torch.cuda.synchronize()
        for _ in range(5):
            for v in vs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    ops_pm.mlp(x, w, b, 1, out=out, tile_hint=v)
                e1.record(); torch.cuda.synchronize()
                tm[v].append(e0.elapsed_time(e1) * 1e3 / 6)
        fl = 2.0 * K * C * R
        print("K=%4d C=%4d |" % (K, C) + " |".join(" %s %7.1f us %.3f" % ("lds" if v == 7 else "T=%d" % (v >> 8), np.median(tm[v]), fl / np.median(tm[v]) / 1e6 / 157.3) for v in vs), flush=True)
