#!/usr/bin/env python3
"""Sweep of the guided schedule of the tile-sequence GEMM form (plan = T_A | T_B << 4 | B tiles / 16 << 8 | C tiles / 16 << 16) on the
seven long-row fp32 launches of a bench step, alone on the chip; prints the LDS-tiled form, the automatic plan and the best plans.
    python scripts/seq_gemm_sweep.py [--rounds 3] [--reps 6]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import ops_pm, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
PEAK = 157.3
SHAPES = [(1024, 0, 2304, 8, 4800, 0, "cnn  z-GEMM 60x80"), (1024, 0, 1024, 8, 4800, 48, "path p2r ds3"), (256, 0, 576, 8, 19200, 0, "cnn  z-GEMM 120x160"),
          (512, 0, 1024, 8, 4800, 0, "cnn  psp bottleneck"), (512, 0, 512, 8, 4800, 192, "path p2r ds2"), (256, 0, 256, 8, 19200, 192, "path p2r up0"),
          (64, 64, 384, 8, 12288, 0, "path heads' first layer")]
lib = _lib.load()
torch.manual_seed(0)
for K1, K2, C, B, P, py, role in SHAPES:
    x1 = torch.randn(B, P, K1, device=dev)
    x2 = torch.randn(B, P, K2, device=dev) if K2 else None
    w = torch.randn(C, K1 + K2, device=dev) / (K1 + K2) ** 0.5
    b = torch.randn(C, device=dev)
    gather = (torch.randn(B, py, C, device=dev), torch.randint(0, py, (B, P), device=dev)) if py else None
    out = torch.empty(B, P, C, device=dev)
    n_ct = (C + 127) // 128
    plans = []
    for ta in (2, 3, 4, 5, 6, 8):
        if ta > n_ct:
            continue
        for tb in (1, 2, 3):
            if tb >= ta:
                continue
            for bt in ((0,) if tb == 1 else (512, 1024, 1536, 2048)):
                for ct in (0, 256, 512, 768, 1024, 1536):
                    plans.append(ta | tb << 4 | (bt // 16) << 8 | (ct // 16) << 16)
    auto = lib.ffb6d_mlp_pm_seq_plan(B * P, C)
    variants = [7, 8 + 256 * auto] + [8 + 256 * pl for pl in plans]
    ref = ops_pm.mlp(x1, w, b, 1, x2=x2, gather=gather, tile_hint=7).clone()
    times = {v: [] for v in variants}
    for v in variants[:8]:
        assert torch.equal(ops_pm.mlp(x1, w, b, 1, x2=x2, gather=gather, out=out, tile_hint=v), ref), (role, v)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops_pm.mlp(x1, w, b, 1, x2=x2, gather=gather, out=out, tile_hint=v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    fl = 2.0 * (K1 + K2) * C * B * P
    med = {v: float(np.median(t)) for v, t in times.items()}
    name = lambda v: "lds" if v == 7 else "T=%d/%d B=%d C=%d" % ((v >> 8) & 15, (v >> 12) & 15, ((v >> 16) & 255) * 16, ((v >> 24) & 127) * 16)
    print("%-26s K=%4d C=%4d rows=%6d tiles=%d" % (role, K1 + K2, C, B * P, n_ct * ((B * P + 127) // 128)))
    print("    lds  %7.1f us %.3f    auto (%s) %7.1f us %.3f" % (med[7], fl / med[7] / 1e6 / PEAK, name(variants[1]), med[variants[1]], fl / med[variants[1]] / 1e6 / PEAK))
    for v in sorted(variants[2:], key=lambda v: med[v])[:6]:
        print("    %-24s %7.1f us %.3f" % (name(v), med[v], fl / med[v] / 1e6 / PEAK), flush=True)
