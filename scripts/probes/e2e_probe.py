#!/usr/bin/env python3
"""Where does an iteration of the overlapped sensor -> pose pipeline spend its time?  Host stamps around the enqueue of the forward, the
input assembly and the pose solver (whose polling blocks the host), GPU stamps (HIP events) of the forward's begin / end on the main stream
and of the pose solver's on its side stream.    python scripts/e2e_probe.py [--iters 6]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from ffb6d_amd import model, pipeline, synth

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
net = model.FFB6D(n_classes=22, n_pts=12288)
net.load_state_dict(bench.state_dict(22))
net = net.to(dev).eval()
net.two_streams = True
B, N = 8, 12288
fr = synth.make_batch(2, B, n_points=N)
sensor = {"rgb": torch.from_numpy(fr["rgb"]).to(dev), "depth": torch.from_numpy(np.ascontiguousarray(fr["dpt_xyz"][:, 2])).to(dev)}
cases = [synth.make_pose_case(900 + b, n_pts=N, n_obj=5, mesh_seed=9) for b in range(B)]
stack = lambda k: torch.from_numpy(np.stack([c[k] for c in cases])).to(dev)
votes = (stack("pcld"), stack("mask"), stack("ctr_of"), stack("kp_of"))
pipe = pipeline.SensorToPose(net, synth.LINEMOD_K, N, cases[0]["mesh_kps"], cases[0]["mesh_ctr"], r_lst=cases[0]["r_lst"],
                             pose_inputs=lambda i, o: votes, seed=7)
pipe.run([sensor] * 3, overlap=True)
torch.cuda.synchronize()
main = torch.cuda.current_stream()
s_in, s_pose = pipe._side()
ev = lambda: torch.cuda.Event(enable_timing=True)
origin = ev(); origin.record(main)
t_origin = time.perf_counter()
rows = []
inp = pipe.assemble(sensor, 0)
pending = None
for n in range(a.iters):
    h0 = time.perf_counter()
    f0, f1 = ev(), ev()
    f0.record(main)
    out = pipe.forward(inp)
    f1.record(main)
    h1 = time.perf_counter()
    with torch.cuda.stream(s_in):
        nxt = pipe.assemble(sensor, n + 1)
    h2 = time.perf_counter()
    p0 = p1 = None
    if pending is not None:
        s_pose.wait_event(pending[2])
        with torch.cuda.stream(s_pose):
            p0, p1 = ev(), ev()
            p0.record(s_pose)
            pipe.solve(pending[0], pending[1])
            p1.record(s_pose)
    h3 = time.perf_counter()
    e = torch.cuda.Event(); e.record(main)
    pending = (inp, out, e)
    main.wait_stream(s_in)
    inp = nxt
    rows.append((h0, h1, h2, h3, f0, f1, p0, p1))
torch.cuda.synchronize()
print("iter | host: forward enqueue, inputs enqueue, pose call (ms) | GPU: forward begin..end, pose begin..end (ms after the origin)")
for n, (h0, h1, h2, h3, f0, f1, p0, p1) in enumerate(rows):
    g = lambda e: origin.elapsed_time(e)
    print("%3d | %6.2f %6.2f %6.2f  (host begin %7.2f) | fwd %7.2f .. %7.2f (%.2f) | pose %s" % (
        n, (h1 - h0) * 1e3, (h2 - h1) * 1e3, (h3 - h2) * 1e3, (h0 - t_origin) * 1e3, g(f0), g(f1), f0.elapsed_time(f1),
        "%7.2f .. %7.2f (%.2f)" % (g(p0), g(p1), p0.elapsed_time(p1)) if p0 is not None else "-"))
