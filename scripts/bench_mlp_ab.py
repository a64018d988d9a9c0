#!/usr/bin/env python3
"""A/B of the two shared-MLP GEMM kernels on the per-frame (MFMA-bound) layer shapes of one FFB6D
forward (bs=8, N=12288): FFB6D_MLP_PIPE=0 (first generation) vs 1 (buffer-load pipelined loop, the default);
`python scripts/bench_mlp_ab.py 1 2` adds the experimental LDS-direct variant (2, Cout > 64 only).
Each variant runs in its own process (the switch is read once).  The max|err| column is against a torch
baddbmm reference, the `sum` column must be identical between variants (same tiles, same summation order)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, K1, K2, Cout, P, gather rows or 0)
SHAPES = [("ds3 p2r_fuse 1024->1024 +gather", 1024, 0, 1024, 4800, 48), ("psp bottleneck 512->1024 +gather", 512, 0, 1024, 4800, 4800),
          ("ds2 p2r_fuse 512->512 +gather", 512, 0, 512, 4800, 192), ("up0 p2r_fuse 256->256 +gather", 256, 0, 256, 19200, 192),
          ("ds1 p2r_fuse 128->128 +gather", 128, 0, 128, 4800, 768), ("ds0 p2r_fuse 64->64 +gather", 64, 0, 64, 19200, 3072),
          ("up1 p2r_fuse 64->64 +gather", 64, 0, 64, 76800, 768), ("head [64;64]->128", 64, 64, 128, 12288, 0),
          ("head 128->128", 128, 0, 128, 12288, 0), ("dec [64;.]->64 +gather", 64, 0, 64, 12288, 3072),
          ("att0 pooled mlp 32->64", 32, 0, 64, 12288, 0), ("res0 [32;16]->64", 32, 16, 64, 12288, 0)]

if os.environ.get("FFB6D_MLP_AB_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    from ffb6d_amd import ops
    dev = torch.device("cuda:0")
    B = 8
    for name, k1, k2, C, P, py in SHAPES[:int(os.environ.get("FFB6D_MLP_AB_N", len(SHAPES)))]:
        torch.manual_seed(0)
        x1 = torch.randn(B, k1, P, device=dev)
        x2 = torch.randn(B, k2, P, device=dev) if k2 else None
        wt = (torch.randn(C, k1 + k2, device=dev) / (k1 + k2) ** 0.5).t().contiguous()
        bias = torch.randn(C, device=dev)
        gather = None
        if py:
            gather = (torch.randn(B, C, py, device=dev), torch.randint(0, py, (B, P), device=dev))
        fn = lambda: ops.shared_mlp(x1, wt, bias, ops.ACT_RELU, x2=x2, gather=gather)
        out = fn()
        ref = torch.baddbmm(bias.view(1, -1, 1), wt.t().unsqueeze(0).expand(B, -1, -1), torch.cat([x1, x2], 1) if k2 else x1)
        if py:
            ref = ref + torch.gather(gather[0], 2, gather[1].unsqueeze(1).expand(-1, C, -1))
        err = float((out - torch.relu(ref)).abs().max())
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print("%-36s %9.1f us %7.1f TF/s  max|err| %.2e  sum %.6e" % (name, us, 2.0 * B * (k1 + k2) * C * P / us / 1e6, err, float(out.double().sum())), flush=True)
else:
    for v in (sys.argv[1:] or ["0", "1"]):
        print("== FFB6D_MLP_PIPE=" + v, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, FFB6D_MLP_PIPE=v, FFB6D_MLP_AB_CHILD="1"), check=False)
