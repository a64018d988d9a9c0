"""Probe: MIOpen fp32 convolutions of the colour branch, NCHW vs channels_last (planning data)."""
import torch
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cases = [("up1 1024->256 3x3 @120x160", 1024, 256, 120, 160, 1), ("layer3 256->256 3x3 d2 @60x80", 256, 256, 60, 80, 2),
         ("layer4 512->512 3x3 d4 @60x80", 512, 512, 60, 80, 4), ("layer1 64->64 3x3 @120x160", 64, 64, 120, 160, 1),
         ("up2 256->64 3x3 @240x320", 256, 64, 240, 320, 1), ("up3 64->64 3x3 @480x640", 64, 64, 480, 640, 1)]
for name, ci, co, h, w, d in cases:
    conv = torch.nn.Conv2d(ci, co, 3, padding=d, dilation=d, bias=False).to(dev)
    x = torch.randn(8, ci, h, w, device=dev)
    res = {}
    for fmt in ("nchw", "nhwc"):
        c, xx = conv, x
        if fmt == "nhwc":
            c = conv.to(memory_format=torch.channels_last); xx = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for _ in range(3): c(xx)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): c(xx)
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        res[fmt] = ms
    fl = 2 * 8 * ci * co * 9 * h * w
    print("%-34s nchw %.3f ms (%.0f TF)  nhwc %.3f ms (%.0f TF)" % (name, res["nchw"], fl / res["nchw"] / 1e9, res["nhwc"], fl / res["nhwc"] / 1e9), flush=True)
