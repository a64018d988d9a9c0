import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffb6d_amd import loss, model as M, pyramid, synth
dev = torch.device("cuda:0")
PER, NPT, H, W, NCLS = 2, 1024, 120, 160, 4
def batch(first, count):
    fr = [synth.make_frame(synth.frame_seed(3, first + s), n_points=NPT, height=H, width=W) for s in range(count)]
    frames = {k: np.stack([f[k] for f in fr]) for k in fr[0]}
    tg = [synth.make_targets(synth.frame_seed(3, first + s), frames["cld"][s], n_classes=NCLS) for s in range(count)]
    t = tuple(torch.from_numpy(np.stack([x[k] for x in tg])).to(dev) for k in ("labels", "kp_targ_ofst", "ctr_targ_ofst"))
    inputs = pyramid.frames_to_device(frames, dev)
    inputs["rgb"] = inputs["rgb"].contiguous(memory_format=torch.channels_last)
    return inputs, (t[0].long(),) + t[1:]
def build(cl=True):
    torch.manual_seed(0)
    net = M.FFB6D(n_classes=NCLS, n_pts=NPT)
    import json
    if os.environ.get("SYNTH"):
        shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
        net.load_state_dict(synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=NCLS))
    net = net.to(dev)
    if cl: net = net.to(memory_format=torch.channels_last)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.dropout._DropoutNd): m.p = 0.0
    return net
names = ["rndla_ds_stages.0.lfa.att_pooling_1.fc.weight", "cnn_pre_stages.1.weight", "ctr_ofst_layer.0.conv.weight"]
def grads_of(module, inputs, targets):
    module.zero_grad()
    out = module(inputs)
    l = loss.training_loss(out, *targets)[0]
    l.backward()
    params = dict(module.named_parameters())
    return float(l), {n: params[n].grad.detach().float().flatten()[:256].cpu().numpy() for n in names}, {k: v.detach().float().cpu().numpy() for k, v in out.items()}
inputs, targets = batch(0, 2)
net = build()
if os.environ.get("EVALBN"):
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.eval()
l1, g1, o1 = grads_of(net, inputs, targets)
l2, g2, o2 = grads_of(net, inputs, targets)
net2 = build()
sd1, sd2 = net.state_dict(), net2.state_dict()
print("weights equal:", all(torch.equal(sd1[k], sd2[k]) for k in sd1 if "running" not in k and "num_batches" not in k))
l3, g3, o3 = grads_of(net2, inputs, targets)
print("loss", l1, l2, l3)
for k in o1: print("out", k, np.abs(o1[k]-o2[k]).max(), np.abs(o1[k]-o3[k]).max(), np.abs(o1[k]).max())
for n in names: print(n, np.abs(g1[n]-g2[n]).max()/np.abs(g1[n]).max(), np.abs(g1[n]-g3[n]).max()/np.abs(g1[n]).max())
