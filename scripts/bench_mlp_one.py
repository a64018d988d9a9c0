#!/usr/bin/env python3
"""Run the largest shared-MLP GEMM of FFB6D (1024->1024 on 4800 px x 8 frames, K=1024 after the p2r
split) a few times -- target for `rocprofv3 --pmc` runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops
dev = torch.device("cuda:0")
K, C, P, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1024, 4800, 8
x = torch.randn(B, K, P, device=dev)
wt = (torch.randn(K, C, device=dev) / K ** 0.5).contiguous()
bias = torch.randn(C, device=dev)
for _ in range(10):
    ops.shared_mlp(x, wt, bias, ops.ACT_RELU)
torch.cuda.synchronize()
