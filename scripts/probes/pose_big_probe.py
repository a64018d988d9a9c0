#!/usr/bin/env python3
"""Sets of more than 4096 points through the pose solver's two forms (csrc/pose.hip: 0 = round-by-round, 1 = chip-wide rounds with duplicate
merging, then the one-workgroup fit): time per call and rounds made.   python scripts/probes/pose_big_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import _lib, pose, synth

dev = torch.device("cuda:0")
lib = _lib.load()
rng = np.random.RandomState(5)
cases = {}
fr = synth.make_batch(2, 8, n_points=12288, height=480, width=640)
scene = np.ascontiguousarray(fr["cld_rgb_nrm"][:, :3].transpose(0, 2, 1)).astype(np.float32)        # [8,12288,3]
cases["8 scene clouds of 12288 points (the e2e bench's network-votes case)"] = scene
obj = np.stack([np.concatenate([0.01 * rng.randn(6000, 3) + [0.1, -0.2, 0.9], 0.02 * rng.randn(2000, 3) + [0.35, 0.1, 1.1]]) for _ in range(8)]).astype(np.float32)
cases["8 object-like vote sets of 8000 points"] = obj
for name, pts in cases.items():
    G, M, _ = pts.shape
    sets = torch.zeros((G, M, 4), device=dev)
    sets[:, :, :3] = torch.from_numpy(pts).to(dev)
    counts = torch.full((G,), M, dtype=torch.int32, device=dev)
    print(name)
    for form in (0, 1):
        lib.ffb6d_pose_set_big_form(form)
        pose.mean_shift(sets, counts, 0.04)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            c, l, n, r = pose.mean_shift(sets, counts, 0.04)
        torch.cuda.synchronize()
        print("   form %d: %8.2f ms per call; rounds %s; ball sizes %s" % (form, 1e3 * (time.perf_counter() - t0) / 3, r.tolist(), n.tolist()))
lib.ffb6d_pose_set_big_form(1)
