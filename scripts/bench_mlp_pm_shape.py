#!/usr/bin/env python3
"""Ten launches of one point-major shared-MLP GEMM shape for PMC collection (scripts/pmc_pm_shape.sh).
argv: K Cout rows [dtype f32|bf16] [tile_hint]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffb6d_amd import ops_pm
dev = torch.device("cuda:0")
K, C, R = (int(v) for v in sys.argv[1:4])
dt = torch.bfloat16 if len(sys.argv) > 4 and sys.argv[4] == "bf16" else torch.float32
hint = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(0)
x = torch.randn(R, K, device=dev).to(dt)
w = (torch.randn(C, K, device=dev) / K ** 0.5).to(dt)
b = torch.randn(C, device=dev)
out = torch.empty(R, C, device=dev, dtype=dt)
for _ in range(10):
    ops_pm.mlp(x, w, b, 1, out=out, tile_hint=hint)
torch.cuda.synchronize()
