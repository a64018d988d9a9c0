import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import test_forward_gpu as T
dev = torch.device("cuda:0")
loss, ours, ref_loss, ref, _ = T._step_gradients(dev, 1024, 120, 160, 2, False)
rows = []
for n, r in ref.items():
    if r is None or float(r.abs().max()) == 0: continue
    g = ours[n]
    rows.append((float((g - r).abs().max()) / float(r.abs().max()), n, float(r.abs().max()), float(g.abs().max())))
rows.sort(reverse=True)
print("loss", loss, ref_loss)
for e, n, sr, sg in rows[:25]:
    print("%.3e  %-60s ref max %.3e ours max %.3e" % (e, n, sr, sg))
print("params with err > 5e-3:", sum(e > 5e-3 for e, *_ in rows), "of", len(rows))
