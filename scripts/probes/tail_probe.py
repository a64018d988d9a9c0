"""Tile-schedule tail probe for csrc/mlp_pm.hip: one 1024->1024 (and 512->512, 256->256) GEMM at row counts that give
2304 / 2400 / 3072 tiles (768 resident workgroups = 3.0 / 3.125 / 4.0 rounds).  Usage: python scripts/tail_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")


def run(rows, k, cout, hint=0, reps=20):
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(cout, k, device=dev) * 0.03
    b = torch.randn(cout, device=dev)
    out = torch.empty(rows, cout, device=dev)
    for _ in range(5):
        ops_pm.mlp(x, w, b, act=1, out=out, tile_hint=hint)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        ops_pm.mlp(x, w, b, act=1, out=out, tile_hint=hint)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    return us, 2.0 * rows * k * cout / us * 1e-6


for k, cout in ((1024, 1024), (512, 512), (256, 256), (128, 128)):
    ct = cout // 64
    for tiles in (768, 1536, 2304, 2400, 2688, 3072, 3840, 4608):
        rows = tiles // ct * 256
        for hint in (2, 1):
            us, tf = run(rows, k, cout, hint)
            print(f"{k}->{cout} rows {rows:7d} tiles(64x256) {rows // 256 * ct:5d} rounds {rows // 256 * ct / 768:5.3f} hint {hint}: {us:8.1f} us {tf:6.1f} TF", flush=True)
