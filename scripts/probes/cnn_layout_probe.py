#!/usr/bin/env python3
"""Planning probe: the colour branch's MIOpen convolutions end to end (stem, 4 encoder stages, 4 decoder stages, no
fusion), NCHW vs channels_last, plain torch modules, bs=8 480x640."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import model
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
net = model.FFB6D(n_classes=22, n_pts=12288).to(dev).eval()
for p in net.parameters():
    p.requires_grad_(False)


def cnn(x):
    y = net.cnn_pre_stages(x)
    for st in net.cnn_ds_stages:
        y = st(y)
    for st in net.cnn_up_stages:
        y = st(y)
    return y


for fmt in ("nchw", "nhwc", "nchw", "nhwc"):
    x = torch.randn(8, 3, 480, 640, device=dev)
    if fmt == "nhwc":
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    else:
        net = net.to(memory_format=torch.contiguous_format)
    with torch.enable_grad():      # grad mode on + frozen parameters = stock torch modules, no autograd graph
        for _ in range(3):
            y = cnn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = cnn(x)
        e1.record()
        torch.cuda.synchronize()
    print("%s: %.2f ms per batch of 8 (out %s, contiguous_cl=%s)" % (
        fmt, e0.elapsed_time(e1) / 5, tuple(y.shape), y.is_contiguous(memory_format=torch.channels_last)), flush=True)
