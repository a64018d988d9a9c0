"""tests/golden/make_golden_f4.py -- committed vectors for SURVEY 8f rank 4 (surface normals, depth hole filling).  normalSpeed and
cv2 are absent from this image, so these vectors come from the restatements oracle/inputs_ref.depth_normal and
oracle/holefill_ref.fill_missing (checked primitive by primitive and against analytic surfaces in tests/test_f4_pin_cpu.py), NOT from
the reference's third-party binaries: "parity unpinned" for this row stays in force.  What they buy: the GPU box compares the HIP
kernels with fixed bytes instead of with whatever the restatement computes there."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
from oracle import holefill_ref, inputs_ref  # noqa: E402
import test_f4_pin_cpu as F  # noqa: E402

out = {}
for cam in ("linemod", "ycb"):
    fx, fy = F.CAMS[cam][:2]
    fxs, fys = fx * 160 / 640, fy * 120 / 480
    for name, z in (("plane", F.plane_depth(cam, np.array([0.3, 0.1, -1.0]) / np.linalg.norm([0.3, 0.1, -1.0]), 900.0)),
                    ("sphere", F.sphere_depth(cam, np.array([20.0, -10.0, 900.0]), 260.0)[0])):
        tag = f"{cam}_{name}"
        rng = np.random.RandomState(len(tag))
        z = z + (z > 0) * rng.normal(0, 1.5, z.shape)                      # sensor-like noise: exercises the difference threshold
        z[rng.rand(*z.shape) < 0.03] = 0
        out[tag + "/depth_mm"] = z.astype(np.float32)
        out[tag + "/fxfy"] = np.array([fxs, fys], np.float64)
        out[tag + "/normals"] = inputs_ref.depth_normal(out[tag + "/depth_mm"], fxs, fys, 5, 2000, 20, False)
for seed in (0, 1):
    d = (F.holes(seed, 96, 128) * 10000.0).astype(np.uint16)
    tag = f"holes{seed}"
    out[tag + "/depth_raw"] = d
    out[tag + "/cam_scale"] = np.float64(10000.0)
    out[tag + "/filled"] = holefill_ref.fill_missing(d, 10000.0, 1)
np.savez_compressed(os.path.join(HERE, "f4_vectors.npz"), **out)
print("f4 vectors written:", sorted(out))
