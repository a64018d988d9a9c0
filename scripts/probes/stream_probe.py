"""A/B of the stream form of the point-major GEMM (tile_hint 6, csrc/mlp_pm.hip mlp_pm_stream_kernel) against the default tile
choice on the HBM-bound layers of the step, fp32 and bf16: time, algorithmic GB/s, and the largest difference of the results.
Usage: python scripts/stream_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")


def run(fn, reps=20):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps, y


shapes = [  # (k1, k2, cout, rows, act, gather?)
    (64, 0, 64, 2457600, 1), (64, 0, 64, 614400, 1), (64, 0, 64, 98304, 1), (64, 0, 64, 24576, 1),
    (32, 0, 32, 393216, 2), (32, 16, 64, 98304, 2), (32, 16, 64, 393216, 2), (64, 64, 128, 98304, 1), (128, 0, 128, 98304, 1),
    (128, 0, 128, 393216, 1), (64, 0, 32, 2457600, 3), (64, 0, 24, 2457600, 3), (256, 0, 128, 98304, 1), (256, 0, 64, 153600, 1), (64, 0, 64, 100000, 0),
]
for dt in (torch.float32, torch.bfloat16):
    esz = 4 if dt == torch.float32 else 2
    for k1, k2, cout, rows, act in shapes:
        if (k1 + k2) * esz > 512 or (k1 + k2) * esz < 96:
            continue
        torch.manual_seed(0)
        x1 = torch.randn(rows, k1, device=dev).to(dt)
        x2 = torch.randn(rows, k2, device=dev).to(dt) if k2 else None
        w = (torch.randn(cout, k1 + k2, device=dev) / (k1 + k2) ** 0.5).to(dt)
        b = torch.randn(cout, device=dev)
        o0 = torch.empty(rows, cout, device=dev, dtype=dt)
        o6 = torch.empty(rows, cout, device=dev, dtype=dt)
        t0, y0 = run(lambda: ops_pm.mlp(x1, w, b, act, x2=x2, out=o0))
        t6, y6 = run(lambda: ops_pm.mlp(x1, w, b, act, x2=x2, out=o6, tile_hint=6))
        by = rows * (k1 + k2 + cout) * esz
        diff = float((y0.float() - y6.float()).abs().max())
        if k2 == 0 and act != 3 and rows % 8 == 0:      # the same layer with a gathered epilogue row (p2r fusion / decoder: W_a x + gather(W_b e))
            Y = torch.randn(8, 3072, cout, device=dev).to(dt)
            gi = torch.randint(0, 3072, (8, rows // 8), device=dev)
            xg = x1.view(8, rows // 8, k1)
            tg, _ = run(lambda: ops_pm.mlp(xg, w, b, act, gather=(Y, gi), tile_hint=6))
            byg = by + rows * 8 + rows * cout * esz
            extra = " | +gather %7.1f us %6.0f GB/s" % (tg, byg / tg * 1e-3)
        else:
            extra = ""
        print(f"{'f32' if esz == 4 else 'bf16'} [{k1}+{k2}]->{cout} rows {rows:8d} act {act}: default {t0:7.1f} us {by / t0 * 1e-3:6.0f} GB/s | "
              f"stream {t6:7.1f} us {by / t6 * 1e-3:6.0f} GB/s | maxdiff {diff:.2e}" + extra, flush=True)
