"""Where the time of mean_shift_fit_kernel goes: one vote set (and 256 copies of it) fitted with the round limit at 1, 2, 3, ... rounds.
    python scripts/pose_fit_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import pose, synth
dev = torch.device("cuda:0")
c = synth.make_pose_case(900, n_pts=12288, n_obj=5, mesh_seed=9)
sel = c["mask"] == np.unique(c["mask"][c["mask"] > 0])[0]
votes = (c["pcld"][sel] - c["kp_of"][0][sel]).astype(np.float32)
M = votes.shape[0]
for G in (1, 256, 320):
    sets = torch.zeros((G, M, 4), device=dev)
    sets[:, :, :3] = torch.from_numpy(votes).to(dev)
    counts = torch.full((G,), M, dtype=torch.int32, device=dev)
    line = "G=%3d M=%d:" % (G, M)
    for it in (0, 1, 2, 3, 4, 6, 8, 16, 32, 64, 128, 300):
        for _ in range(2):
            pose.mean_shift(sets, counts, 0.05, it, want_labels=False, check_every=0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _, _, _, r = pose.mean_shift(sets, counts, 0.05, it, want_labels=False, check_every=0)
        e1.record(); torch.cuda.synchronize()
        line += " %d:%.0fus(r%d)" % (it + 1, e0.elapsed_time(e1) * 1e3 / 3, int(r[0]))
    print(line, flush=True)
