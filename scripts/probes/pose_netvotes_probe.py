#!/usr/bin/env python3
"""What the random-init network's own votes look like to the pose solver (the e2e bench's network-votes variant): class histogram of the
argmax mask, size of the offsets, and the mean-shift calls on them in both large-set forms.   python scripts/probes/pose_netvotes_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from ffb6d_amd import _lib, distributed, model, pipeline, pose, synth

dev = torch.device("cuda:0")
lib = _lib.load()
torch.backends.cudnn.benchmark = True
net = model.FFB6D(n_classes=22, n_pts=12288)
net.load_state_dict(bench.state_dict(22))
net = net.to(dev).eval()
net.two_streams, net.precision, net.index_dtype = True, "fp32", torch.int64
frames = distributed.shard_frames(2, 8, 0, None, n_points=12288)
sensor = {"rgb": torch.from_numpy(frames["rgb"]).to(dev), "depth": torch.from_numpy(np.ascontiguousarray(frames["dpt_xyz"][:, 2])).to(dev)}
rng = np.random.RandomState(3)
pipe = pipeline.SensorToPose(net, synth.LINEMOD_K, 12288, ((rng.rand(22, 8, 3) - 0.5) * 0.2).astype(np.float32),
                             ((rng.rand(22, 3) - 0.5) * 0.02).astype(np.float32), r_lst=(0.08 + 0.05 * rng.rand(21)).astype(np.float32))
inp = pipe.assemble(sensor, 0)
out = pipe.forward(inp)
mask = out["pred_rgbd_segs"].argmax(dim=1)
print("classes per frame:", [torch.bincount(m, minlength=22).tolist() for m in mask[:3]])
print("|ctr offsets| mean %.3f max %.3f   |kp offsets| mean %.3f max %.3f" % (out["pred_ctr_ofs"].norm(dim=-1).mean(), out["pred_ctr_ofs"].norm(dim=-1).max(),
                                                                               out["pred_kp_ofs"].norm(dim=-1).mean(), out["pred_kp_ofs"].norm(dim=-1).max()))
B = mask.shape[0]
pairs = [(b, int(c)) for b in range(B) for c in torch.unique(mask[b]).tolist() if c > 0]
frame_of = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
class_of = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
for name, off, spc in (("centre votes", out["pred_ctr_ofs"], 1), ("keypoint votes", out["pred_kp_ofs"], 8)):
    sets, counts = pose.vote_sets(inp["cld"], off, mask, frame_of, class_of)
    print(name, "sets", sets.shape[0], "counts", counts.tolist())
    ext = sets[0, :int(counts[0]), :3]
    print("   extent of set 0:", (ext.max(0)[0] - ext.min(0)[0]).tolist())
    for form in (0, 1):
        lib.ffb6d_pose_set_big_form(form)
        pose.mean_shift(sets, counts, 0.04, sets_per_count=spc, want_labels=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c, l, n, r = pose.mean_shift(sets, counts, 0.04, sets_per_count=spc, want_labels=False)
        torch.cuda.synchronize()
        print("   form %d: %8.2f ms; rounds %s; ball sizes %s" % (form, 1e3 * (time.perf_counter() - t0), r.tolist()[:16], n.tolist()[:16]))
lib.ffb6d_pose_set_big_form(1)
