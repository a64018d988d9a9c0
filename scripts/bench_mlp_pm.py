#!/usr/bin/env python3
"""Point-major shared MLP (csrc/mlp_pm.hip) against the channel-major kernel (csrc/shared_mlp.hip) on the layer shapes
of one FFB6D forward (bs=8, N=12288): correctness vs a float64 reference and time per launch.  `hint` = tile_hint."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops, ops_pm

dev = torch.device("cuda:0")
B = 8
# (name, K1, K2, Cout, P per frame, gather rows per frame or 0)
SHAPES = [("ds3 p2r_fuse 1024->1024 +gather", 1024, 0, 1024, 4800, 48), ("psp bottleneck 512->1024 +add", 512, 0, 1024, 4800, -1),
          ("ds2 p2r_fuse 512->512 +gather", 512, 0, 512, 4800, 192), ("up0 p2r_fuse 256->256 +gather", 256, 0, 256, 19200, 192),
          ("ds1 p2r_fuse 128->128 +gather", 128, 0, 128, 4800, 768), ("ds0 p2r_fuse 64->64 +gather", 64, 0, 64, 19200, 3072),
          ("up1 p2r_fuse 64->64 +gather", 64, 0, 64, 76800, 768), ("final 64->64 @307200", 64, 0, 64, 307200, 0),
          ("head [64;64]->128", 64, 64, 128, 12288, 0), ("head 128->128", 128, 0, 128, 12288, 0), ("head 128->22", 128, 0, 22, 12288, 0),
          ("dec [64;.]->64 +gather", 64, 0, 64, 12288, 3072), ("att0 pooled mlp 32->64", 32, 0, 64, 12288, 0),
          ("res0 [32;16]->64", 32, 16, 64, 12288, 0), ("lfa0 mlp1 16->16 pairs", 16, 0, 16, 196608, 0),
          ("lfa1 mlp2 32->32 pairs", 32, 0, 32, 49152, 0), ("res2 [128;128]->256 @768", 128, 128, 256, 768, 0),
          ("ds3 y 1024->1024 @48", 1024, 0, 1024, 48, 0), ("ds3 r2p_fuse [512;512]->512 @48", 512, 512, 512, 48, 0),
          ("res3 [256;256]->512 @192", 256, 256, 512, 192, 0), ("dec0 768->256 @192", 512, 256, 256, 192, 0)]
hints = [int(h) for h in sys.argv[1:]] or [0]
if os.environ.get("MLP_PM_BIG"):
    SHAPES = SHAPES[:4] + SHAPES[8:10]
if os.environ.get("MLP_PM_STREAM"):
    SHAPES = [s for s in SHAPES if s[1] + s[2] <= 64 and s[4] >= 12288]


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


for name, k1, k2, C, P, py in SHAPES:
    torch.manual_seed(0)
    K = k1 + k2
    x1 = torch.randn(B, P, k1, device=dev)
    x2 = torch.randn(B, P, k2, device=dev) if k2 else None
    w = torch.randn(C, K, device=dev) / K ** 0.5
    bias = torch.randn(C, device=dev)
    gather = add = None
    if py > 0:
        gather = (torch.randn(B, py, C, device=dev), torch.randint(0, py, (B, P), device=dev))
    elif py < 0:
        add = torch.randn(B, P, C, device=dev)
    X = torch.cat([x1, x2], 2) if k2 else x1
    ref = X.double() @ w.double().t() + bias.double()
    if gather:
        ref = ref + torch.gather(gather[0].double(), 1, gather[1].unsqueeze(2).expand(-1, -1, C))
    if add is not None:
        ref = ref + add.double()
    ref = torch.relu(ref)
    scale = float(ref.abs().max())
    line = "%-34s" % name
    for h in hints:
        try:
            out = ops_pm.mlp(x1, w, bias, ops.ACT_RELU, x2=x2, add=add, gather=gather, tile_hint=h)
            err = float((out.double() - ref).abs().max()) / scale
            us = timeit(lambda: ops_pm.mlp(x1, w, bias, ops.ACT_RELU, x2=x2, add=add, gather=gather, tile_hint=h))
            line += "  pm[h=%d] %8.1f us %6.1f TF %6.0f GB/s err %.1e" % (
                h, us, 2.0 * B * K * C * P / us / 1e6, 4.0 * B * P * (K + C) / us / 1e3, err)
        except Exception as e:  # noqa: BLE001
            line += "  pm[h=%d] FAILED %s" % (h, str(e)[:60])
    # channel-major kernel on the transposed problem
    if C >= 8 and P % 4 == 0 and (k2 == 0 or k1 % 16 == 0):
        c1 = x1.transpose(1, 2).contiguous()
        c2 = x2.transpose(1, 2).contiguous() if k2 else None
        g = None
        if gather:
            g = (gather[0].transpose(1, 2).contiguous(), gather[1])
        elif add is not None:
            g = (add.transpose(1, 2).contiguous(), torch.arange(P, device=dev, dtype=torch.int32).repeat(B, 1))
        wt = w.t().contiguous()
        us = timeit(lambda: ops.shared_mlp(c1, wt, bias, ops.ACT_RELU, x2=c2, gather=g))
        line += "  | cm %8.1f us %6.1f TF" % (us, 2.0 * B * K * C * P / us / 1e6)
    print(line, flush=True)
