#!/usr/bin/env python3
"""One launch per half of the local feature aggregation (csrc/lfa_pm.hip) against the round-2 chain it replaces
(posenc_mlp -> att_pool -> mlp [-> mlp -> att_pool -> mlp]) on the four level shapes of BASELINE configuration 2 (bs=8,
N=12288: 12288 / 3072 / 768 / 192 points per frame, d = 32 / 64 / 128 / 256), fp32 and bf16, both point-group sizes.
Prints time per launch, algorithmic TFLOP/s (scores + mlp1 [+ mlp2] + output MLP) and GB/s at the fused boundary
(xyz + indices + point rows in, point rows out), and the largest difference between the two paths.
    python scripts/bench_lfa.py [N0=12288] [idx_bits=64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")
B = 8
N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
IDT = torch.int32 if len(sys.argv) > 2 and sys.argv[2] == "32" else torch.int64


def timeit(fn, n=20):
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


for dt in (torch.float32, torch.bfloat16):
    esz = 4 if dt == torch.float32 else 2
    for lvl, d in enumerate((32, 64, 128, 256)):
        N, h = N0 >> (2 * lvl), d // 2
        g = torch.Generator().manual_seed(d)
        xyz = ops_pm.xyz_table(torch.rand(B, N, 3, generator=g).to(dev))
        # neighbours as the real pyramid has them: nearby in space, scattered in memory
        nei = torch.randint(0, N, (B, N, 16), generator=g).to(IDT).to(dev)
        f = torch.randn(B, N, h, generator=g).to(dt).to(dev)
        w1, b1 = (torch.randn(h, 10, generator=g) / 2).to(dev), (torch.randn(h, generator=g) / 2).to(dev)
        w2, b2 = (torch.randn(h, h, generator=g) / h ** 0.5).to(dt).to(dev), (torch.randn(h, generator=g) / 2).to(dev)
        wfc = [(torch.randn(d, d, generator=g) / d ** 0.5 * 2).to(dt).to(dev) for _ in range(2)]
        wm = [(torch.randn(c, d, generator=g) / d ** 0.5).to(dt).to(dev) for c in (h, d)]
        bm = [(torch.randn(c, generator=g) / 2).to(dev) for c in (h, d)]

        def half1(p_hint=0):
            return ops_pm.lfa_half(1, xyz, nei, f, w1, b1, 2, wfc[0], wm[0], bm[0], 2, p_hint=p_hint)

        agg = half1()

        def half2(p_hint=0):
            return ops_pm.lfa_half(2, xyz, nei, agg, w1, b1, 2, wfc[1], wm[1], bm[1], 2, w2=w2, b2=b2, act2=2, p_hint=p_hint)

        def chain1():
            g1 = ops_pm.posenc_mlp(xyz[..., :3].contiguous(), nei, w1, b1, 2, dtype=dt)
            return ops_pm.mlp(ops_pm.att_pool(f, nei, g1, wfc[0]), wm[0], bm[0], 2), g1

        def chain2(g1):
            return ops_pm.mlp(ops_pm.att_pool(agg, nei, ops_pm.mlp(g1, w2, b2, 2), wfc[1]), wm[1], bm[1], 2)

        want1, g1 = chain1()
        want2 = chain2(g1)
        got2 = half2()
        e1 = float((agg.float() - want1.float()).abs().max()) / float(want1.float().abs().max())
        e2 = float((got2.float() - want2.float()).abs().max()) / float(want2.float().abs().max())
        ib = 8 if IDT == torch.int64 else 4
        for mode, fn, ref in ((1, half1, lambda: chain1()), (2, half2, lambda: chain2(g1))):
            cout = h if mode == 1 else d
            flops = 2 * 16 * B * N * (d * d + 10 * h + (h * h if mode == 2 else 0)) + 2 * B * N * d * cout
            nbytes = B * N * (12 + 16 * ib + esz * (h + cout))
            t_ref = timeit(ref)
            line = "%s L%d d=%3d N=%5d half %d: chain %7.1f us |" % ("f32 " if esz == 4 else "bf16", lvl, d, N, mode, t_ref)
            for ph in ((2, 3, 4, 10) if d <= 64 else (1, 2)):          # 4: one wave per workgroup; + 8: weights resident in LDS
                t = timeit(lambda: fn(ph))
                line += " P=%2d%s %6.1f us %5.1f TF %4.0f GB/s |" % (128 // d if ph & 7 == 4 else 1024 // d >> ((ph & 7) - 1), "w1" if ph & 7 == 4 else ("LW" if ph & 8 else "  "), t, flops / t * 1e-6, nbytes / t * 1e-3)
            print(line + " maxdiff vs chain %.1e" % (e1 if mode == 1 else e2), flush=True)
