"""The index pyramid of one batch, N times (for rocprofv3 passes over the exact-KNN kernels):  python scripts/pyramid_loop.py [--iters 6]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffb6d_amd import pyramid, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--n-points", type=int, default=12288)
ap.add_argument("--iters", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
frames = synth.make_batch(2, a.batch, n_points=a.n_points)
cld = torch.from_numpy(frames["cld"]).to(dev)
dpt = torch.from_numpy(frames["dpt_xyz"]).to(dev)
for _ in range(30):          # the caching allocator needs a few rounds to settle (single calls of 8-36 ms among the first twenty)
    pyramid.build_index_pyramid(cld, dpt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    pyramid.build_index_pyramid(cld, dpt)
e1.record()
torch.cuda.synchronize()
print("build_index_pyramid: %.1f us per batch of %d frames" % (e0.elapsed_time(e1) * 1e3 / a.iters, a.batch))
