#!/usr/bin/env python3
"""Which tensors does the training step still copy / cast, and where from?  (needs a GPU)

    python scripts/train_copy_census.py [--precision bf16|fp32] [--batch 8] > gpurun_out/train_copy_census.txt

One bf16-autocast training step of bench.py's configuration under torch.profiler (CPU + device activities, shapes, Python
stacks): every aten::copy_ / _to_copy / contiguous / clone / cat with its input shape, dtype, device time and the innermost
frame of this package that issued it (blank = issued from autograd's backward).  profiles/r03_rocprofv3_kernel_stats_train_
bf16_rows.txt still lists 44 strided fp32 copies per step (3.6 ms) that the kernel trace cannot attribute."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile

from ffb6d_amd import model, pyramid, synth

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--n-points", type=int, default=12288)
args = ap.parse_args()

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
frames = synth.make_batch(2, args.batch, n_points=args.n_points)
net = model.FFB6D(n_classes=22, n_pts=args.n_points).to(dev).train().to(memory_format=torch.channels_last)
opt = torch.optim.Adam(net.parameters(), lr=1e-5)
inputs = pyramid.frames_to_device(frames, dev)
inputs["rgb"] = inputs["rgb"].float().contiguous(memory_format=torch.channels_last)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16"):
        out = net(inputs)
        loss = sum((v.float() ** 2).mean() for v in out.values())
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

WATCH = ("aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::cat")
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name not in WATCH:
        continue
    shapes = [tuple(s) for s in (e.input_shapes or []) if s]
    site = next((s for s in (e.stack or []) if "ffb6d_amd/" in s), "")
    key = (e.name, str(shapes[:2]), site.split("ffb6d_amd/")[-1][:60])
    rows[key][0] += 1
    rows[key][1] += getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
print(f"# one training step, {args.precision}, bs={args.batch}: copy-like ATen ops by (op, input shapes, call site); device time in us")
for key, (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:80]:
    print(f"{us:10.1f} us  x{n:<4d} {key[0]:18s} {key[1]:60s} {key[2]}")
