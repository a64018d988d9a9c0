"""The two GEMMs of the step's tail (profiles/r05_design_md_history_notes.md 4b'): K = 576 -> 64 on the picked pixels' patch rows and the heads' stacked first layer
(64 + 64 -> 384, two sources), per kernel form (tile_hint), fp32 at bs = 8 and bf16 at bs = 16.  Usage: python scripts/tail_gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ffb6d_amd import ops_pm

dev = torch.device("cuda:0")


def run(fn, reps=20):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps, y


torch.manual_seed(0)
for dt, bs in ((torch.float32, 8), (torch.bfloat16, 16)):
    rows = 12288 * bs
    for k1, k2, cout in ((576, 0, 64), (64, 64, 384), (128, 0, 128)):
        x1 = torch.randn(rows, k1, device=dev).to(dt)
        x2 = torch.randn(rows, k2, device=dev).to(dt) if k2 else None
        w = (torch.randn(cout, k1 + k2, device=dev) / (k1 + k2) ** 0.5).to(dt)
        b = torch.randn(cout, device=dev)
        base = None
        line = []
        for hint in (0, 7, 1, 2, 4):
            try:
                us, y = run(lambda: ops_pm.mlp(x1, w, b, 1, x2=x2, tile_hint=hint))
            except Exception as e:      # a form that does not take the shape
                line.append(f"{hint}: -")
                continue
            base = y if base is None else base
            line.append(f"{hint}: {us:7.1f} us {2e-6 * rows * (k1 + k2) * cout / us:6.1f} TF/s{'' if torch.equal(y, base) else ' (bits differ)'}")
        print(f"{str(dt)[6:]:9s} K={k1}+{k2} cout={cout} rows={rows}   " + "   ".join(line))

# a head's three layers after the first (128 -> 128 -> 128 -> c): three launches against csrc/mlp_chain.hip, fp32, bs = 8
rows = 12288 * 8
wide = torch.randn(rows, 384, device=dev)
x = wide[:, 128:256]
for cout in (22, 24, 3):
    ws = [torch.randn(128, 128, device=dev) / 11, torch.randn(128, 128, device=dev) / 11, torch.randn(cout, 128, device=dev) / 11]
    bs = [torch.randn(128, device=dev), torch.randn(128, device=dev), torch.randn(cout, device=dev)]
    cpad = -(-cout // 4) * 4
    wp, bp = torch.zeros(cpad, 128, device=dev), torch.zeros(cpad, device=dev)
    wp[:cout], bp[:cout] = ws[2], bs[2]
    w32, b32 = torch.zeros(32, 128, device=dev), torch.zeros(32, device=dev)
    w32[:cout], b32[:cout] = ws[2], bs[2]
    parts = [(ops_pm.k_chunked(ws[0]), bs[0], 1), (ops_pm.k_chunked(ws[1]), bs[1], 1), (ops_pm.k_chunked(w32), b32, 0)]
    t3, y3 = run(lambda: ops_pm.mlp(ops_pm.mlp(ops_pm.mlp(x, ws[0], bs[0], 1), ws[1], bs[1], 1), wp, bp, 0))
    t1, y1 = run(lambda: ops_pm.mlp_chain3(x, parts[0], parts[1], parts[2], cpad))
    fl = 2e-6 * rows * (2 * 128 * 128 + 128 * cout)
    print(f"head chain 128->128->128->{cout}: three launches {t3:7.1f} us   one launch {t1:7.1f} us {fl / t1:6.1f} TF/s   "
          f"max |diff| / range {float((y1 - y3).abs().max() / y3.abs().max()):.1e}")
