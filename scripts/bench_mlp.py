#!/usr/bin/env python3
"""Micro-benchmark: fused shared-MLP MFMA kernel vs torch (rocBLAS/hipBLASLt baddbmm + activation)
on the GEMM shapes of one FFB6D forward (bs=8, N=12288).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops

dev = torch.device("cuda:0")
B = 8
# (name, K, Cout, P)
shapes = [("fc0 9->8", 9, 8, 12288), ("lfa0.mlp1 10->16", 10, 16, 12288 * 16), ("att0.fc 32x32", 32, 32, 12288 * 16),
          ("att1.fc 64x64", 64, 64, 3072 * 16), ("att2.fc 128", 128, 128, 768 * 16), ("att3.fc 256", 256, 256, 192 * 16),
          ("ds0 p2r_fuse 128->64", 128, 64, 19200), ("ds1 p2r_fuse 256->128", 256, 128, 4800),
          ("ds2 p2r_fuse 1024->512", 1024, 512, 4800), ("ds3 p2r_fuse 2048->1024", 2048, 1024, 4800),
          ("up0 p2r_fuse 512->256", 512, 256, 19200), ("up1 p2r_fuse 128->64", 128, 64, 76800),
          ("deep 1024->512 P48", 1024, 512, 48), ("deep 256->256 P192", 256, 256, 192), ("deep 512->256 P192", 512, 256, 192),
          ("deep 1024->1024 P48", 1024, 1024, 48), ("deep 128->128 P768", 128, 128, 768),
          ("head 128->128", 128, 128, 12288), ("res0 [32;8]->64", 40, 64, 12288), ("res3 [256;256]->512", 512, 512, 192)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-28s %10s %10s %10s %10s" % ("shape", "ours us", "TF/s", "torch us", "TF/s"))
for name, K, C, P in shapes:
    x = torch.randn(B, K, P, device=dev)
    w = torch.randn(C, K, device=dev) / K ** 0.5
    wt = w.t().contiguous()
    bias = torch.randn(C, device=dev)
    ours = timeit(lambda: ops.shared_mlp(x, wt, bias, ops.ACT_RELU))
    ref = timeit(lambda: torch.relu_(torch.baddbmm(bias.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), x)))
    fl = 2.0 * B * K * C * P
    print("%-28s %10.1f %10.1f %10.1f %10.1f" % (name, ours, fl / ours / 1e6, ref, fl / ref / 1e6))
