#!/usr/bin/env python3
"""One half of the fused local feature aggregation (csrc/lfa_pm.hip) on one level shape, N launches -- the workload of
scripts/pmc_lfa.sh.   python scripts/bench_lfa_one.py LEVEL(0-3) MODE(1|2) [f32|bf16] [p_hint] [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops_pm

lvl, mode = int(sys.argv[1]), int(sys.argv[2])
dt = torch.bfloat16 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else torch.float32
p_hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
B, d = 8, 32 << lvl
N, h = 12288 >> (2 * lvl), d // 2
cout = h if mode == 1 else d
g = torch.Generator().manual_seed(d)
xyz = ops_pm.xyz_table(torch.rand(B, N, 3, generator=g).to(dev))
nei = torch.randint(0, N, (B, N, 16), generator=g).to(dev)
f = torch.randn(B, N, h, generator=g).to(dt).to(dev)
w1, b1 = (torch.randn(h, 10, generator=g) / 2).to(dev), (torch.randn(h, generator=g) / 2).to(dev)
w2, b2 = (torch.randn(h, h, generator=g) / h ** 0.5).to(dt).to(dev), (torch.randn(h, generator=g) / 2).to(dev)
wfc = (torch.randn(d, d, generator=g) / d ** 0.5 * 2).to(dt).to(dev)
wm, bm = (torch.randn(cout, d, generator=g) / d ** 0.5).to(dt).to(dev), (torch.randn(cout, generator=g) / 2).to(dev)
kw = dict(w2=w2, b2=b2, act2=2) if mode == 2 else {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n + 2):
    if i == 2:
        e0.record()
    ops_pm.lfa_half(mode, xyz, nei, f, w1, b1, 2, wfc, wm, bm, 2, p_hint=p_hint, **kw)
e1.record()
torch.cuda.synchronize()
flops = 2 * 16 * B * N * (d * d + 10 * h + (h * h if mode == 2 else 0)) + 2 * B * N * d * cout
t = e0.elapsed_time(e1) / n * 1e3
print("L%d half %d %s p_hint %d: %.1f us  %.1f TFLOP/s" % (lvl, mode, "bf16" if dt == torch.bfloat16 else "f32", p_hint, t, flops / t * 1e-6))
