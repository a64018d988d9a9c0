#!/usr/bin/env python3
"""A/B of the GEMM forms on the seven long-row fp32 launches of a bench step, alone on the chip: the LDS-tiled form (hint 7) against the
tile-sequence form (hint 8 + 256 T) for several T, interleaved in one process (rounds x variants x launches), median per launch.
    python scripts/seq_gemm_probe.py [--rounds 5] [--reps 8]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import ops_pm

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--ts", default="2,3,4,6,8")
a = ap.parse_args()
dev = torch.device("cuda:0")
PEAK = 157.3
# (K1, K2, Cout, frames, rows per frame, gathered Y rows per frame or 0, role)
SHAPES = [(1024, 0, 2304, 8, 4800, 0, "cnn  z-GEMM 60x80"), (1024, 0, 1024, 8, 4800, 48, "path p2r ds3"), (256, 0, 576, 8, 19200, 0, "cnn  z-GEMM 120x160"),
          (512, 0, 1024, 8, 4800, 0, "cnn  psp bottleneck"), (512, 0, 512, 8, 4800, 192, "path p2r ds2"), (256, 0, 256, 8, 19200, 192, "path p2r up0"),
          (64, 64, 384, 8, 12288, 0, "path heads' first layer")]
ts = [int(t) for t in a.ts.split(",")]
name = lambda v: "auto" if v == 0 else "auto(r05 plans)" if v == -1 else "lds " if v == 7 else ("lin%d" % ((v >> 8) & 15) if ((v >> 12) & 15) == 15 else "T=%d" % (v >> 8))
torch.manual_seed(0)
tot = {}
for K1, K2, C, B, P, py, role in SHAPES:
    x1 = torch.randn(B, P, K1, device=dev)
    x2 = torch.randn(B, P, K2, device=dev) if K2 else None
    w = torch.randn(C, K1 + K2, device=dev) / (K1 + K2) ** 0.5
    b = torch.randn(C, device=dev)
    gather = (torch.randn(B, py, C, device=dev), torch.randint(0, py, (B, P), device=dev)) if py else None
    out = torch.empty(B, P, C, device=dev)
    n_ct = (C + 127) // 128
    variants = [7] + [8 + 256 * t for t in ts if t <= n_ct] + ([8 + 256 * (0xF0 | r) for r in (1, 2, 3, 4)] if n_ct <= 8 else []) + [0, -1]
    ref = ops_pm.mlp(x1, w, b, 1, x2=x2, gather=gather, tile_hint=7).clone()
    times = {v: [] for v in variants}
    lib = ops_pm._lib.load()

    def run(v):
        lib.ffb6d_mlp_pm_set_seq_lin(0 if v == -1 else 1)
        return ops_pm.mlp(x1, w, b, 1, x2=x2, gather=gather, out=out, tile_hint=max(v, 0))
    for v in variants:
        got = run(v)
        assert torch.equal(got, ref), (role, v)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    fl = 2.0 * (K1 + K2) * C * B * P
    line = "%-26s K=%4d C=%4d rows=%6d |" % (role, K1 + K2, C, B * P)
    for v in variants:
        us = float(np.median(times[v]))
        tot.setdefault(v, 0.0)
        tot[v] += us
        line += " %s %7.1f us %.3f |" % (name(v), us, fl / us / 1e6 / PEAK)
    print(line, flush=True)
print("sum over the shapes a variant ran on:", {name(v): round(t, 1) for v, t in tot.items()})
