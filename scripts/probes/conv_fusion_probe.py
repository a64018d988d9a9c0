#!/usr/bin/env python3
"""Planning probe: MIOpen's fused conv+bias+ReLU (torch.miopen_convolution_relu / _add_relu) against conv followed by the
affine_act_pm pass, on the colour branch's 3x3 shapes, channels_last, fp32 and bf16 (bs=8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from ffb6d_amd import ops, ops_pm
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


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


for dt in (torch.float32, torch.bfloat16):
    for name, ci, co, h, w in [("layer1 64->64 @120x160", 64, 64, 120, 160), ("layer2 128->128 @60x80", 128, 128, 60, 80),
                               ("layer3 256->256 @60x80", 256, 256, 60, 80), ("layer4 512->512 @60x80", 512, 512, 60, 80),
                               ("up1 1024->256 @120x160", 1024, 256, 120, 160), ("up3 64->64 @480x640", 64, 64, 480, 640)]:
        x = torch.randn(8, ci, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        scale, shift = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
        wf = (wt.float() * scale.view(-1, 1, 1, 1)).to(dt).contiguous(memory_format=torch.channels_last)
        bias = shift.to(dt)
        res = torch.randn(8, co, h, w, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)

        def unfused():
            y = F.conv2d(x, wt, None, 1, 1).permute(0, 2, 3, 1)
            return ops_pm.affine_act_(y, scale, shift, act=ops.ACT_RELU)

        def fused():
            return torch.miopen_convolution_relu(x, wf, bias, [1, 1], [1, 1], [1, 1], 1)

        def unfused_res():
            y = F.conv2d(x, wt, None, 1, 1).permute(0, 2, 3, 1)
            return ops_pm.affine_act_(y, scale, shift, act=ops.ACT_RELU, residual=res.permute(0, 2, 3, 1))

        def fused_res():
            return torch.miopen_convolution_add_relu(x, wf, res, 1.0, bias, [1, 1], [1, 1], [1, 1], 1)

        line = "%-8s %-26s" % (str(dt).split(".")[1], name)
        for label, fn in (("conv+affine", unfused), ("fused", fused), ("conv+affine+res", unfused_res), ("fused+res", fused_res)):
            try:
                line += "  %s %7.1f us" % (label, timeit(fn))
            except Exception as e:  # noqa: BLE001
                line += "  %s FAILED(%s)" % (label, str(e)[:40])
        try:
            a, b = unfused().permute(0, 3, 1, 2).float(), fused().float()
            line += "  maxdiff %.2e cl=%s" % (float((a - b).abs().max()), fused().is_contiguous(memory_format=torch.channels_last))
        except Exception as e:  # noqa: BLE001
            line += "  cmp FAILED"
        print(line, flush=True)
