import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_forward_gpu as T
dev = torch.device("cuda:0")
loss, ours, ref_loss, ref, (plain_loss, plain) = T._step_gradients(dev, 1100, 136, 168, 2, True)
print("loss ours", loss, "fp32 ref", ref_loss, "plain bf16", plain_loss)
rows = []
for n, r in ref.items():
    if r is None or float(r.abs().max()) == 0: continue
    nr = float(r.double().norm())
    rows.append((float((ours[n].double() - r.double()).norm()) / nr, float((plain[n].double() - r.double()).norm()) / nr, n))
rows.sort(key=lambda t: t[0] - 2 * t[1], reverse=True)
for eo, ep, n in rows[:20]: print("%.3e %.3e %s" % (eo, ep, n))
import numpy as np
print("median ours %.3e plain %.3e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
