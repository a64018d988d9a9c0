"""A/B of the LDS-tiled form of the point-major GEMM (tile_hint 7, mlp_pm_lds_kernel) against the default choice on the long-row
layers, bf16 and fp32: time, TFLOP/s, largest difference of the results.  Usage: python scripts/lds_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")


def run(fn, reps=10):
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


# (k1, k2, cout, rows): p2r fusion GEMMs ds3 / ds2 / up0, PSP bottleneck, r2p fuse, head layers -- at bs=16 (bf16) / bs=8 (fp32)
shapes = [(1024, 0, 1024, 4800), (512, 0, 1024, 4800), (512, 0, 512, 4800), (256, 0, 256, 19200), (128, 0, 256, 19200),
          (512, 256, 256, 192), (256, 0, 128, 12288), (64, 64, 128, 12288), (128, 0, 128, 12288)]
for dt, bs in ((torch.bfloat16, 16), (torch.float32, 8)):
    for k1, k2, cout, per in shapes:
        rows = per * bs
        torch.manual_seed(0)
        x1 = torch.randn(rows, k1, device=dev).to(dt)
        x2 = torch.randn(rows, k2, device=dev).to(dt) if k2 else None
        w = (torch.randn(cout, k1 + k2, device=dev) / (k1 + k2) ** 0.5).to(dt)
        b = torch.randn(cout, device=dev)
        o0 = torch.empty(rows, cout, device=dev, dtype=dt)
        o7 = torch.empty(rows, cout, device=dev, dtype=dt)
        t0, y0 = run(lambda: ops_pm.mlp(x1, w, b, 1, x2=x2, out=o0))
        t7, y7 = run(lambda: ops_pm.mlp(x1, w, b, 1, x2=x2, out=o7, tile_hint=7))
        fl = 2.0 * rows * (k1 + k2) * cout
        print(f"{'bf16' if dt == torch.bfloat16 else 'f32 '} [{k1}+{k2}]->{cout} rows {rows:7d}: default {t0:7.1f} us {fl / t0 * 1e-6:6.0f} TF | "
              f"lds {t7:7.1f} us {fl / t7 * 1e-6:6.0f} TF | maxdiff {float((y0.float() - y7.float()).abs().max()):.2e}", flush=True)
