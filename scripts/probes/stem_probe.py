"""Colour stem (7x7 stride-2 convolution 3 -> 64 on 480x640, pspnet/extractors.py:104) on MIOpen in four formulations:
the plain channels-last call, input channels zero-padded to 4 / 8, and space-to-depth (stride-1 4x4 convolution on 12
channels).  Prints time per call and the max deviation from the plain result.  Usage: python scripts/stem_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
B = 8
x = torch.rand(B, 3, 480, 640, device=dev) * 255
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05


def timeit(fn, reps=20):
    for _ in range(5):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, y


xcl = x.contiguous(memory_format=torch.channels_last)
wcl = w.contiguous(memory_format=torch.channels_last)
t, ref = timeit(lambda: F.conv2d(xcl, wcl, None, 2, 3))
print("plain NHWC C=3          %.3f ms" % t, ref.shape, ref.is_contiguous(memory_format=torch.channels_last))
t, y = timeit(lambda: F.conv2d(x, w, None, 2, 3))
print("plain NCHW C=3          %.3f ms  maxdiff %.3e" % (t, float((y - ref).abs().max())))
for c in (4, 8):
    xp = F.pad(x, (0, 0, 0, 0, 0, c - 3)).contiguous(memory_format=torch.channels_last)
    wp = F.pad(w, (0, 0, 0, 0, 0, c - 3)).contiguous(memory_format=torch.channels_last)
    t, y = timeit(lambda: F.conv2d(xp, wp, None, 2, 3))
    print("channels padded to %d     %.3f ms  maxdiff %.3e (range %.3e)" % (c, t, float((y - ref).abs().max()), float(ref.abs().max())))


def s2d_input(x):
    xp = F.pad(x, (3, 5, 3, 5))                                   # 486+2 x 646+2 -> 488 x 648
    Bn, C, H, W = xp.shape
    xs = xp.view(Bn, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(Bn, 4 * C, H // 2, W // 2)
    return xs.contiguous(memory_format=torch.channels_last)


w8 = F.pad(w, (0, 1, 0, 1))                                        # 8x8, zero last row / column
ws = w8.view(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 4, 4).contiguous(memory_format=torch.channels_last)
t_in, xs = timeit(lambda: s2d_input(x))
t, y = timeit(lambda: F.conv2d(xs, ws, None, 1, 0))
print("space-to-depth: input rearrangement %.3f ms, 4x4 conv on 12 ch %.3f ms, out %s" % (t_in, t, tuple(y.shape)))
y = y[:, :, :240, :320]
print("   maxdiff %.3e (range %.3e)" % (float((y - ref).abs().max()), float(ref.abs().max())))
for c in (16,):
    xs16 = F.pad(xs, (0, 0, 0, 0, 0, c - 12)).contiguous(memory_format=torch.channels_last)
    ws16 = F.pad(ws, (0, 0, 0, 0, 0, c - 12)).contiguous(memory_format=torch.channels_last)
    t, y = timeit(lambda: F.conv2d(xs16, ws16, None, 1, 0))
    print("space-to-depth padded to 16 ch: %.3f ms maxdiff %.3e" % (t, float((y[:, :, :240, :320] - ref).abs().max())))
