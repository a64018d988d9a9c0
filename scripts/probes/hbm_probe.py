#!/usr/bin/env python3
"""Planning probe (one MI355X): what the memory system gives for the access patterns the gather kernels can
choose between -- pure writes, streaming copy, channel-major 4-byte gathers (today's layout) and point/pixel-major
row gathers (a gathered element = C contiguous floats).  Prints achieved GB/s per pattern."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ffb6d_amd import ops

dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (40, 157, 630, 2500):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev)
    b = torch.randn(n, device=dev)
    t = timeit(lambda: a.fill_(1.0))
    print("fill   %5d MB: %7.1f us  %6.0f GB/s (write only)" % (mb, t * 1e6, 4 * n / t / 1e9))
    t = timeit(lambda: a.copy_(b))
    print("copy   %5d MB: %7.1f us  %6.0f GB/s (read+write)" % (mb, t * 1e6, 8 * n / t / 1e9))
    t = timeit(lambda: torch.add(b, 1.0, out=a))
    print("add    %5d MB: %7.1f us  %6.0f GB/s (read+write)" % (mb, t * 1e6, 8 * n / t / 1e9))
    del a, b

B = 8
# (name, C, M, U): nearest_interpolation shapes of the forward (channel-major 4-byte gathers)
for name, C, M, U in [("p2r up1 64ch 768->76800", 64, 768, 76800), ("p2r up0 256ch 192->19200", 256, 192, 19200),
                      ("choose 64ch 307200->12288", 64, 307200, 12288), ("lfa0 16ch 12288->196608", 16, 12288, 196608),
                      ("dec3 64ch 3072->12288", 64, 3072, 12288)]:
    f = torch.randn(B, C, M, device=dev)
    idx = torch.randint(0, M, (B, U, 1), device=dev)
    t = timeit(lambda: ops.nearest_interpolation(f, idx))
    nb = 4 * B * C * M + 8 * B * U + 4 * B * C * U
    print("nearest_interp cm  %-28s %7.1f us  %6.0f GB/s" % (name, t * 1e6, nb / t / 1e9))
    # point-major rows: out[b,u,:] = f_pm[b,idx,:]  (C contiguous floats per gathered element)
    fpm = f.transpose(1, 2).contiguous()
    flat = (idx.view(B, U) + torch.arange(B, device=dev).view(B, 1) * M).view(-1)
    src = fpm.view(B * M, C)
    out = torch.empty(B * U, C, device=dev)
    t = timeit(lambda: torch.index_select(src, 0, flat, out=out))
    print("index_select rows  %-28s %7.1f us  %6.0f GB/s" % (name, t * 1e6, nb / t / 1e9))

for name, C, M, Np in [("r2p up2 64ch 76800->3072", 64, 76800, 3072), ("r2p ds0 64ch 19200->3072", 64, 19200, 3072),
                       ("pool0 64ch 12288->3072", 64, 12288, 3072), ("r2p ds3 1024ch 4800->48", 1024, 4800, 48),
                       ("r2p up0 256ch 19200->192", 256, 19200, 192)]:
    f = torch.randn(B, C, M, device=dev)
    idx = torch.randint(0, M, (B, Np, 16), device=dev)
    t = timeit(lambda: ops.random_sample(f, idx))
    nb = 4 * B * C * M + 8 * B * Np * 16 + 4 * B * C * Np
    print("random_sample cm   %-28s %7.1f us  %6.0f GB/s" % (name, t * 1e6, nb / t / 1e9))
