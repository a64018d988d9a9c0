#!/usr/bin/env python3
"""Time profile of the one-workgroup mean-shift fit (csrc/pose.hip: mean_shift_fit_kernel) over its rounds: the same vote sets with
max_iter = 0, 1, 2, 4, 8, 16, 50, 100, 300 (HIP events around the launch alone).   python scripts/pose_rounds_probe.py [--sets 40|320]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import pose, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
N = 12288
cases = [synth.make_pose_case(900 + b, n_pts=N, n_obj=5, mesh_seed=9) for b in range(a.batch)]
stack = lambda k: torch.from_numpy(np.stack([c[k] for c in cases])).to(dev)
pcld, mask, ctr_of, kp_of = stack("pcld"), stack("mask"), stack("ctr_of"), stack("kp_of")
frame_of = torch.arange(a.batch, device=dev, dtype=torch.int32).repeat_interleave(5)
class_of = torch.arange(1, 6, device=dev, dtype=torch.int32).repeat(a.batch)
for name, off, spc in (("centre votes, 40 sets", ctr_of, 1), ("keypoint votes, 320 sets", kp_of, 8)):
    sets, counts = pose.vote_sets(pcld, off, mask, frame_of, class_of)
    print(name, "mean points", float(counts.float().mean()))
    for it in (0, 1, 2, 4, 8, 16, 50, 100, 300):
        pose.mean_shift(sets, counts, 0.04, it, sets_per_count=spc, want_labels=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _, _, _, rounds = pose.mean_shift(sets, counts, 0.04, it, sets_per_count=spc, want_labels=False)
        e1.record()
        torch.cuda.synchronize()
        print("   max_iter %3d: %8.1f us per call, rounds made: mean %.1f max %d" % (it, e0.elapsed_time(e1) * 1e3 / 3, float(rounds.float().mean()), int(rounds.max())))
