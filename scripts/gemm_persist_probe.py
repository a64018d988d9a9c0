"""A/B of the point-major LDS-tiled GEMM forms on the launches of the benchmarked step (bs = 8, N = 12288, fp32; bf16 with --bf16):
tile_hint 7 (one tile per workgroup) against 8 (persistent workgroups, operand stream across tiles) and its walk variants (hint bits
8..15 = workgroups per XCD, 16..23 = channel tiles per block).  Interleaved rounds in ONE process, median and minimum per form;
operands like the network's (post-ReLU: half zeros) unless --randn.  Usage: python scripts/gemm_persist_probe.py [--bf16] [--randn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import statistics
import torch

from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")
BF = "--bf16" in sys.argv
RANDN = "--randn" in sys.argv
dt = torch.bfloat16 if BF else torch.float32
# (rows, k1, k2, cout, gathered epilogue rows per frame, choose gather, name)
SHAPES = [(38400, 1024, 0, 2304, 0, False, "z0 1024->2304"), (38400, 1024, 0, 1024, 48, False, "ds3 p2r 1024->1024 +g"),
          (614400, 64, 0, 576, 0, False, "z2 64->576"), (153600, 256, 0, 576, 0, False, "z1 256->576"),
          (38400, 512, 0, 1024, 0, False, "psp 512->1024"), (38400, 512, 0, 512, 192, False, "ds2 p2r 512->512 +g"),
          (153600, 256, 0, 256, 192, False, "up0 p2r 256->256 +g"), (98304, 64, 64, 128, 0, True, "head 128->128 choose"),
          (98304, 128, 0, 128, 0, False, "head 128->128")]
if BF:
    SHAPES = [(2 * r, k1, k2, c, py, xg, n) for r, k1, k2, c, py, xg, n in SHAPES]
FORMS = [("7", 7), ("8", 8)]
if "--epi" in sys.argv:        # probe forms of the persistent kernel's epilogue (hint bits 24..27): fragment-shaped stores, none at all
    FORMS += [("8 frag", 8 + (4 << 24)), ("8 noepi", 8 + (8 << 24))]
if "--spread" in sys.argv:     # start of the workgroups spread over 8 k / 16 k / 32 k cycles (at most one tile period), row and fragment epilogues
    FORMS += [("8 s8k", 8 + (1 << 24)), ("8 s16k", 8 + (2 << 24)), ("8 s32k", 8 + (3 << 24)), ("8 frag s16k", 8 + (6 << 24)), ("8 frag s32k", 8 + (7 << 24)),
              ("8 noepi", 8 + (8 << 24))]
if "--walk" in sys.argv:
    FORMS += [("8 gc=n_ct", 8 + (64 << 16)), ("8 gc=4", 8 + (4 << 16)), ("8 gc=2", 8 + (2 << 16)), ("8 spx=32", 8 + (32 << 8)), ("8 spx=48", 8 + (48 << 8))]
ROUNDS, REPS = 5, 6


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / REPS


print(f"# {'bf16' if BF else 'fp32'} operands {'randn' if RANDN else 'relu(randn)'}; us = median (min) of {ROUNDS} interleaved rounds x {REPS} launches")
tot = {n: 0.0 for n, _ in FORMS}
for rows, k1, k2, cout, py, xg, name in SHAPES:
    torch.manual_seed(0)
    B = 8 * (2 if BF else 1)
    P = rows // B
    act = (lambda t: t) if RANDN else torch.relu
    src = act(torch.randn(B, 2 * P if xg else P, k1, device=dev)).to(dt)
    pick = torch.randint(0, 2 * P, (B, P), device=dev) if xg else None
    x2 = act(torch.randn(B, P, k2, device=dev)).to(dt) if k2 else None
    w = (torch.randn(cout, k1 + k2, device=dev) / (k1 + k2) ** 0.5).to(dt)
    bias = torch.randn(cout, device=dev)
    gather = (torch.randn(B, py, cout, device=dev).to(dt), torch.randint(0, py, (B, P), device=dev)) if py else None
    outs = {n: torch.empty(B, P, cout, device=dev, dtype=dt) for n, _ in FORMS}
    fns = {n: (lambda n=n, h=h: ops_pm.mlp(src, w, bias, 1, x2=x2, gather=gather, x1_gather=pick, out=outs[n], tile_hint=h)) for n, h in FORMS}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    ts = {n: [] for n, _ in FORMS}
    for _ in range(ROUNDS):
        for n, _h in FORMS:
            ts[n].append(timed(fns[n]))
    fl = 2.0 * rows * (k1 + k2) * cout
    line = f"{name:26s} rows {rows:7d}:"
    for n, _h in FORMS:
        med, mn = statistics.median(ts[n]), min(ts[n])
        tot[n] += med
        same = torch.equal(outs[n], outs["7"])
        line += f" | {n} {med:7.1f} ({mn:7.1f}) us {fl / med * 1e-6:5.0f} TF{'' if same else ' DIFF'}"
    print(line, flush=True)
print("sum of medians (us): " + " | ".join(f"{n} {v:8.1f}" for n, v in tot.items()))
