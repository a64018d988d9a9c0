"""Probe: host enqueue time vs GPU time of one step, and hipGraph replay of pyramid+forward."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffb6d_amd import model as M, pyramid, synth

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B, N = 8, 12288
frames = synth.make_batch(1, B, n_points=N, height=480, width=640)
net = M.FFB6D(n_classes=22, n_pts=N)
net.load_state_dict(synth.synth_state_dict(net, seed=0))
net = net.to(dev).eval()
base = pyramid.frames_to_device(frames, dev, with_pyramid=False)
cld = torch.from_numpy(frames['cld']).to(dev)
xyz = torch.from_numpy(frames['dpt_xyz']).to(dev)

def step():
    inp = dict(base)
    inp.update(pyramid.build_index_pyramid(cld, xyz))
    return net(inp)

with torch.no_grad():
    for _ in range(5):
        out = step()
    torch.cuda.synchronize()
    for two in (False, True):
        net.two_streams = two
        for _ in range(3): step()
        torch.cuda.synchronize()
        host = []; tot = []
        for _ in range(10):
            t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            host.append(t1 - t0); tot.append(t2 - t0)
        print("two_streams", two, "host enqueue ms %.2f  total ms %.2f" % (1e3 * min(host), 1e3 * min(tot)), flush=True)
    # graph capture
    for two in (False, True):
        net.two_streams = two
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2): step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                gout = step()
            torch.cuda.synchronize()
            ref = step()
            g.replay(); torch.cuda.synchronize()
            ok = all(torch.equal(gout[k], ref[k]) for k in ref)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print("graph two_streams", two, "equal", ok, "replay ms %.2f" % (1e3 * min(ts)), flush=True)
        except Exception as e:
            print("graph two_streams", two, "FAILED", repr(e)[:400], flush=True)
            torch.cuda.synchronize()
