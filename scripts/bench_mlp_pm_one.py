#!/usr/bin/env python3
"""Ten launches of the largest point-major shared-MLP GEMM of the forward (ds3 p2r_fuse: 1024 -> 1024 on 8 x 4800 pixels,
gather epilogue) for PMC collection (scripts/pmc_pm_gemm.sh).  argv[1] = tile_hint (default 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffb6d_amd import ops, ops_pm
dev = torch.device("cuda:0")
hint = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(0)
x = torch.randn(8, 4800, 1024, device=dev)
w = torch.randn(1024, 1024, device=dev) / 32
b = torch.randn(1024, device=dev)
g = (torch.randn(8, 48, 1024, device=dev), torch.randint(0, 48, (8, 4800), device=dev))
for _ in range(10):
    ops_pm.mlp(x, w, b, ops.ACT_RELU, gather=g, tile_hint=hint)
torch.cuda.synchronize()
