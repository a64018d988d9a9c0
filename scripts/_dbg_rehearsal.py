import os, sys, json, subprocess, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_multigpu_gpu as T
tmp = pathlib.Path(tempfile.mkdtemp())
res = T._run_worker(tmp, 1)
for name, w in res[0]["want"].items():
    w = np.array(w); g = np.array(res[0]["got"][name])
    print("%-55s scale %.3e  max err %.3e  rel %.3e" % (name, np.abs(w).max(), np.abs(w - g).max(), np.abs(w - g).max() / np.abs(w).max()))
for name, (mean, var) in res[0]["want_stats"].items():
    gm, gv = np.array(res[0]["stats"][name][0]), np.array(res[0]["stats"][name][1])
    print(name, "mean err", np.abs(gm - np.array(mean)).max(), "var relerr", (np.abs(gv - np.array(var)) / np.abs(np.array(var))).max())
