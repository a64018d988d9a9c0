"""How long does the host take to enqueue one forward (pyramid streamed inside)?  If that is close to the GPU step time the
step is launch-bound and a captured HIP graph is the fix.  Usage: python scripts/cpu_enqueue_probe.py [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ffb6d_amd import model, synth

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
bs = 16 if bf16 else 8
frames = synth.make_batch(0, bs, n_points=12288, height=480, width=640)
net = model.FFB6D(n_classes=22, n_pts=12288).to(dev).eval()
if bf16:
    net.precision = "bf16"
inputs = {"rgb": torch.from_numpy(frames["rgb"]).to(dev).float(), "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]).to(dev),
          "choose": torch.from_numpy(frames["choose"]).to(dev).long(), "dpt_xyz": torch.from_numpy(frames["dpt_xyz"]).to(dev)}
with torch.no_grad():
    for _ in range(6):
        net(inputs)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        net(inputs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("single step: enqueue %.2f ms, until the GPU is done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        net(inputs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d steps back to back: enqueue %.2f ms/step, GPU %.2f ms/step (bs=%d, %s)" % (n, (t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n, bs,
                                                                                      "bf16" if bf16 else "fp32"))
