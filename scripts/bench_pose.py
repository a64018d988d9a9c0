"""Pose-solver timing on the GPU box: ffb6d_amd.pose.solve_poses on a batch of full-size frames
(HIP events) and rounds made.  (The reference algorithm on the host -- oracle/pose_ref.py under torch with 32
threads -- took 60-100 s per frame of this workload when timed once in round 1; the oracle is test
infrastructure and is not imported here.)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffb6d_amd import _lib, pose, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--n-points", type=int, default=12288)
ap.add_argument("--objects", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--fit-form", type=int, default=1, help="0: the round-5 one-workgroup fit for every set; 1: light form for sets <= 2048 points")
ap.add_argument("--fit-spread", type=int, default=1, help="rounds 0 and 1 of the sets of >= 512 points made chip-wide before the fits: 1 when G <= CUs / 2, 2 always, 0 never")
args = ap.parse_args()

dev = torch.device("cuda:0")
_lib.load().ffb6d_pose_set_fit_form(args.fit_form)
_lib.load().ffb6d_pose_set_fit_spread(args.fit_spread)
cases = [synth.make_pose_case(900 + b, n_pts=args.n_points, n_obj=args.objects, mesh_seed=9) for b in range(args.batch)]
stack = lambda key: torch.from_numpy(np.stack([c[key] for c in cases])).to(dev)
pcld, mask, ctr_of, kp_of = stack("pcld"), stack("mask"), stack("ctr_of"), stack("kp_of")
mk, mc, rl = cases[0]["mesh_kps"], cases[0]["mesh_ctr"], cases[0]["r_lst"]

for _ in range(2):
    res = pose.solve_poses(pcld, mask, ctr_of, kp_of, mk, mc, r_lst=rl)
torch.cuda.synchronize()
tracer = _lib.Tracer()
t0 = time.perf_counter()
_lib.TRACER = tracer
for _ in range(args.steps):
    res = pose.solve_poses(pcld, mask, ctr_of, kp_of, mk, mc, r_lst=rl)
torch.cuda.synchronize()
_lib.TRACER = None
stats = {}
pose.solve_poses(pcld, mask, ctr_of, kp_of, mk, mc, r_lst=rl, stats=stats)
wall = (time.perf_counter() - t0) / args.steps
err = max(np.abs(T - cases[b]["RT"][c]).max() for b, (ids, poses, _) in enumerate(res) for c, T in zip(ids, poses))
out = {"fit_form": args.fit_form, "fit_spread": args.fit_spread, "batch": args.batch, "n_points": args.n_points, "objects_per_frame": args.objects,
       "wall_ms_per_batch": 1e3 * wall, "frames_per_s": args.batch / wall, "max_pose_err_vs_truth": float(err),
       "kernels_ms_per_batch": {k: v["total_ms"] / args.steps for k, v in tracer.summary().items()}}
for k in ("refine", "ctr", "kps"):
    r, c = stats["rounds_" + k].cpu().numpy(), stats["counts_" + k].cpu().numpy()
    out["rounds_" + k] = {"sets": int(r.size), "min": int(r.min()), "mean": float(r.mean()), "max": int(r.max()),
                          "mean_points": float(c.mean())}
print(json.dumps(out))
