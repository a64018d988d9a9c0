"""Debug aid, not a test (GPU box; lives under tests/ because it mixes oracle operators into the model): run ffb6d_amd.model.FFB6D at full size with (a) HIP ops + folded GEMMs,
(b) plain-torch ops (oracle/ops_ref) + folded GEMMs, (c) HIP ops + unfused conv/BN, and print the
per-module max deviation relative to the module's output range, to localise a mismatch."""
import json, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffb6d_amd import model as M, pyramid, synth
from oracle import ops_ref

dev = torch.device("cuda:0")
shapes = json.load(open(os.path.join(ROOT, "tests/golden/state_dict_keys.json")))
net = M.FFB6D(22, 12288)
net.load_state_dict(synth.synth_state_dict_from_shapes(shapes, 0, 22))
net = net.to(dev).eval()
frames = synth.make_batch(1, 1, n_points=12288)
inputs = pyramid.frames_to_device(frames, dev)

def run(mode):
    outs = {}
    hooks = []
    for name, mod in net.named_modules():
        if name and name.count(".") <= 2 and not name.startswith("cnn_"):
            hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: outs.__setitem__(name, o.detach().clone()) if torch.is_tensor(o) else None))
    real_ops = M.ops
    if mode == "torch_ops":
        ns = types.SimpleNamespace(random_sample=ops_ref.random_sample, nearest_interpolation=ops_ref.nearest_interpolation,
                                   gather_neighbour=ops_ref.gather_neighbour, relative_pos_encoding=ops_ref.relative_pos_encoding,
                                   att_pool=ops_ref.att_pool,
                                   choose_gather=lambda r, c: ops_ref.nearest_interpolation(r.reshape(r.shape[0], r.shape[1], -1, 1), c.reshape(c.shape[0], -1, 1)).squeeze(3))
        M.ops = ns
    try:
        if mode == "unfused":
            with torch.enable_grad():
                ep = net(inputs)
        else:
            with torch.no_grad():
                ep = net(inputs)
    finally:
        M.ops = real_ops
        for h in hooks: h.remove()
    outs.update({k: v.detach() for k, v in ep.items()})
    return outs

a = run("hip"); a2 = run("hip"); b = run("torch_ops"); c = run("unfused")
print("%-45s %12s %12s %12s %10s" % ("module", "hip-vs-hip", "hip-vs-torch", "fold-vs-unf", "absmax"))
for k in a:
    s = float(a[k].abs().max()) + 1e-30
    f = lambda x, y: float((x - y).abs().max()) / s
    print("%-45s %12.3e %12.3e %12.3e %10.3e" % (k, f(a[k], a2[k]), f(a[k], b[k]), f(a[k], c[k]), s))
