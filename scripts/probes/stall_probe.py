"""Where do the two streams of the fused forward wait for each other?  Event pairs around every cross-stream wait
(forward_pm.handover), averaged over steps of the default workload (bs=8, N=12288).  Usage: python scripts/stall_probe.py [bf16]   (bf16 = BASELINE config 5: bs=16, bf16 rows)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from ffb6d_amd import model, synth

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
frames = synth.make_batch(0, 16 if bf16 else 8, n_points=12288, height=480, width=640)
net = model.FFB6D(n_classes=22, n_pts=12288).to(dev).eval()
if bf16:
    net.precision = "bf16"
inputs = {"rgb": torch.from_numpy(frames["rgb"]).to(dev).float(), "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]).to(dev),
          "choose": torch.from_numpy(frames["choose"]).to(dev).long(), "dpt_xyz": torch.from_numpy(frames["dpt_xyz"]).to(dev)}
with torch.no_grad():
    for _ in range(6):
        net(inputs)
    torch.cuda.synchronize()
    acc = {}
    total = []
    for _ in range(10):
        net._stall_probe = []
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        net(inputs)
        t1.record()
        torch.cuda.synchronize()
        total.append(t0.elapsed_time(t1))
        for tag, a, b in net._stall_probe:
            acc.setdefault(tag, []).append((t0.elapsed_time(a), a.elapsed_time(b)))
print("step %.2f ms" % np.mean(total))
for tag, v in acc.items():
    v = np.array(v)
    print("%-44s reached at %6.2f ms, waited %5.2f ms" % (tag, v[:, 0].mean(), v[:, 1].mean()))
