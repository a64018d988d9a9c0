#!/usr/bin/env python3
"""A/B of the bf16 long-row GEMM forms of BASELINE configuration 5 (bs = 16), alone on the chip: the LDS-tiled 128 x 128 form (hint 7)
against the 256 x 256 LDS-DMA form (hint 9, csrc/mlp_pm_big.hip), interleaved in one process, median per launch; every variant's
output is compared bit for bit with hint 7's first.
    python scripts/big_gemm_probe.py [--rounds 5] [--reps 8] [--hints 7,9]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import ops_pm

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--hints", default="7,9")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--xpad", type=int, default=0, help="allocate X rows this many elements longer than K (row stride != K: do the rows' segments rotate over the L2 channels?)")
a = ap.parse_args()
dev = torch.device("cuda:0")
PEAK = 2500.0
B = a.batch
# (K1, K2, Cout, rows per frame, gathered Y rows per frame or 0 / -1 = added rows, role)
SHAPES = [(1024, 0, 2304, 4800, 0, "cnn  z-GEMM 60x80 (fold)"), (1024, 0, 1024, 4800, 48, "path p2r ds3"),
          (512, 0, 1024, 4800, -1, "cnn  psp bottleneck"), (256, 0, 576, 19200, 0, "cnn  z-GEMM 120x160 (fold)"),
          (512, 0, 512, 4800, 192, "path p2r ds2"), (256, 0, 256, 19200, 192, "path p2r up0"),
          (64, 64, 384, 12288, 0, "path heads' first layer"), (576, 0, 64, 12288, 0, "cnn  last stage at chosen")]
hints = [int(h) for h in a.hints.split(",")]
torch.manual_seed(0)
BF = torch.bfloat16
tot = {}
for K1, K2, C, P, py, role in SHAPES:
    x1 = torch.randn(B, P, K1 + a.xpad, device=dev).to(BF)[..., :K1]
    x2 = torch.randn(B, P, K2, device=dev).to(BF) if K2 else None
    w = (torch.randn(C, K1 + K2, device=dev) / (K1 + K2) ** 0.5).to(BF)
    b = torch.randn(C, device=dev)
    kw = {}
    if py > 0:
        kw["gather"] = (torch.randn(B, py, C, device=dev).to(BF), torch.randint(0, py, (B, P), device=dev))
    elif py < 0:
        kw["add"] = torch.randn(B, P, C, device=dev).to(BF)
    out = torch.empty(B, P, C, device=dev, dtype=BF)
    ref = ops_pm.mlp(x1, w, b, 1, x2=x2, tile_hint=7, **kw).clone()
    auto = ops_pm._lib.load().ffb6d_mlp_pm_choice(B * P, C, K1, K2, 1, 1, 0) & 255
    variants = []
    for v in hints:
        out.zero_()
        try:
            got = ops_pm.mlp(x1, w, b, 1, x2=x2, out=out, tile_hint=v, **kw)
        except Exception as e:           # a form that does not take this shape
            continue
        if not (v >> 12) & 2:            # (probe variants without an epilogue write nothing; variant = bits 12.. of the hint)
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (role, v)
        variants.append(v)
    times = {v: [] for v in variants}
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops_pm.mlp(x1, w, b, 1, x2=x2, out=out, tile_hint=v, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    fl = 2.0 * (K1 + K2) * C * B * P
    byt = 2.0 * (B * P * (K1 + K2 + C) + C * (K1 + K2)) + (2.0 * B * P * C if py else 0)
    line = "%-28s K=%4d C=%4d rows=%7d auto=%d |" % (role, K1 + K2, C, B * P, auto)
    for v in variants:
        us = float(np.median(times[v]))
        tot.setdefault(v, 0.0)
        tot[v] += us
        line += " hint %d %7.1f us %6.0f TF %.3f  %4.2f TB/s |" % (v, us, fl / us / 1e6, fl / us / 1e6 / PEAK, byt / us / 1e6)
    print(line, flush=True)
print("sum over the shapes:", {v: round(t, 1) for v, t in tot.items()})
