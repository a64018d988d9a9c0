#!/usr/bin/env python3
"""upconv_combine (second half of the folded PSPUpsample, csrc/upconv.hip) alone on the chip: the two maps of a step in fp32 (bs = 8) and
bf16 (bs = 16); bf16 with the 2 x 4 block form on half units (round 6) and with the one-pixel-per-thread form; bit-equality of the two.
    python scripts/upconv_probe.py [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ffb6d_amd import _lib, ops_pm

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default=None, help="f32|bf16,C,form: one configuration (for scripts/pmc_cmd.sh)")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
only = a.only.split(",") if a.only else None
for dt, B in ((torch.float32, 8), (torch.bfloat16, 16)):
    for C, h, w in ((256, 60, 80), (64, 120, 160)):
        if only and (only[0] != ("bf16" if dt == torch.bfloat16 else "f32") or int(only[1]) != C):
            continue
        z = torch.randn(B, h, w, 9 * C, device=dev).to(dt)
        shift = torch.randn(C, device=dev)
        nbytes = z.numel() * z.element_size() + B * 4 * h * w * C * z.element_size()
        res = {}
        outs = {}
        for block in ((int(only[2]),) if only else (3, 1, 0) if dt == torch.bfloat16 else (3, 1)):
            lib.ffb6d_upconv_set_form(block)
            out = ops_pm.upconv_combine(z, shift, 0.25, (2 * h, 2 * w))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops_pm.upconv_combine(z, shift, 0.25, (2 * h, 2 * w))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            res[block] = us
            outs[block] = out
        lib.ffb6d_upconv_set_form(2)
        for k in outs:
            assert only or torch.equal(outs[k].view(torch.uint8), outs[1].view(torch.uint8)), k
        print("%-9s B=%2d C=%3d %3dx%3d -> %3dx%3d  %6.1f MB | " % (str(dt).split(".")[1], B, C, h, w, 2 * h, 2 * w, nbytes / 1e6) +
              " | ".join("%s %7.1f us %5.2f TB/s" % ({3: "lds", 2: "auto", 1: "block", 0: "pixel"}[k], v, nbytes / v / 1e6) for k, v in res.items()), flush=True)
