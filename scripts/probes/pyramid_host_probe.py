"""Host-side timing of the index pyramid call by call (allocator warm-up: single calls of 8-36 ms among the first twenty, 0.32 ms after)
and where torch reads the environment per call:  python scripts/pyramid_host_probe.py"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ffb6d_amd import inputs, synth, pyramid
dev = torch.device("cuda:0")
fr = synth.make_batch(2, 8, n_points=12288)
cld = torch.from_numpy(fr["cld"]).to(dev); dpt = torch.from_numpy(fr["dpt_xyz"]).to(dev)
orig = os._Environ.__getitem__
seen = {}
def spy(self, key):
    t0 = time.perf_counter()
    try:
        return orig(self, key)
    finally:
        dt = time.perf_counter() - t0
        if key not in seen:
            print("ENV LOOKUP", key); traceback.print_stack(limit=7)
        seen.setdefault(key, []).append(dt)
os._Environ.__getitem__ = spy
def both():
    b = pyramid.PyramidBuilder(cld, dpt); b.search_batch(True); b.search_batch(False)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    per = []
    for _ in range(20):
        t1 = time.perf_counter(); both(); per.append((time.perf_counter() - t1) * 1e6)
    print("rep", rep, "host us/iter", [int(x) for x in per])
    print({k: (len(v), int(1e6 * sum(v) / len(v))) for k, v in seen.items()})
