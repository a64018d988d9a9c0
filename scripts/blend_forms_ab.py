"""Tap blend of the folded up-convolution (csrc/upconv.hip) on the three PSPUpsample maps, one form per process:
    FFB6D_UPCONV_COMBINE=simple|select|static python scripts/blend_forms_ab.py
prints microseconds and algorithmic GB/s per map (bs=8) and the checksum of the output (the forms are bit-identical)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ffb6d_amd import ops_pm  # noqa: E402

dev = torch.device("cuda", 0)
form = os.environ.get("FFB6D_UPCONV_COMBINE", "static")
for B, h, w, C in ((8, 60, 80, 256), (8, 120, 160, 64), (8, 240, 320, 64)):
    g = torch.Generator().manual_seed(C + h)
    z = torch.randn(B, h, w, 9 * C, generator=g).to(dev)
    shift = torch.randn(C, generator=g).to(dev)
    out = ops_pm.upconv_combine(z, shift, 0.25, (2 * h, 2 * w))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        out = ops_pm.upconv_combine(z, shift, 0.25, (2 * h, 2 * w))
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 5
    nbytes = 4 * (z.numel() + out.numel())
    print("%-6s %3d ch %3dx%3d -> %3dx%3d  %8.1f us  %7.1f GB/s  checksum %.6f" %
          (form, C, h, w, 2 * h, 2 * w, us, nbytes / (us * 1e-6) / 1e9, float(out.double().sum())), flush=True)
