"""tests/golden/make_golden.py -- regenerates the committed golden fixtures by RUNNING THE
REFERENCE ITSELF (build container only: needs /root/reference and oracle/_ref).

    python tests/golden/make_golden.py [knn] [ops] [model] [forward]

The reference ships no golden vectors or known-answer tests for this path (SURVEY.md
section 4 / 8c), so these files are what pins parity:

  knn_small.npz              small seeded KNN cases: inputs + the reference kd-tree's output
  knn_pyramid_hashes.json    sha256 of each of the 26 index tensors the reference's KNN
                             produces for full-size synthetic frames (N=12288 and N=24576)
  ops_small.npz              the reference's random_sample / nearest_interpolation /
                             gather_neighbour / relative_pos_encoding / Att_pooling outputs
  state_dict_keys.json       parameter/buffer names and shapes of the reference FFB6D
  forward_small.npz          end_points of the reference FFB6D.forward on a small frame
  forward_full_sample.npz    strided samples of end_points at 480x640, N=12288
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ffb6d_amd import synth  # noqa: E402
from oracle import knn as oknn  # noqa: E402
from oracle import pyramid as opyr  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def ref_knn_search(s, q, k):
    """DataProcessing.knn_search of the reference (helper_tool.py:160-170)."""
    _, _, helper_tool = rh.reference_modules()
    return helper_tool.DataProcessing.knn_search(s, q, k)


# (name, seed, B, S, Q, K, kind)
KNN_SMALL_CASES = [
    ("uniform_k16", 11, 2, 500, 300, 16, "uniform"),
    ("uniform_k1", 12, 1, 64, 100, 1, "uniform"),
    ("uniform_k5", 13, 3, 1000, 77, 5, "uniform"),
    ("self_k16", 14, 1, 2500, 2500, 16, "self"),
    ("tiny_k16", 15, 2, 16, 9, 16, "uniform"),
    ("cloud_k32", 16, 1, 1111, 333, 32, "uniform"),
    ("ragged_k3", 17, 1, 1025, 257, 3, "uniform"),
]


def knn_case_inputs(seed, B, S, Q, kind):
    rng = np.random.RandomState(seed)
    sup = rng.rand(B, S, 3).astype(np.float32)
    qry = sup.copy() if kind == "self" else rng.rand(B, Q, 3).astype(np.float32)
    return sup, qry


def make_knn():
    out = {}
    for name, seed, B, S, Q, K, kind in KNN_SMALL_CASES:
        sup, qry = knn_case_inputs(seed, B, S, Q, kind)
        idx = rh.ref_knn_batch(sup, qry, K, omp=True)
        out[name + "/support"] = sup
        out[name + "/query"] = qry
        out[name + "/K"] = np.int64(K)
        out[name + "/idx"] = idx.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "knn_small.npz"), **out)

    hashes = {}
    for tag, config, sample, n_points in (("c2_s0_n12288", 2, 0, 12288), ("c2_s1_n12288", 2, 1, 12288),
                                          ("c4_s0_n24576", 4, 0, 24576),
                                          # the reference's own default: common.py:61 n_sample_points = 12800 ->
                                          # 12800 / 3200 / 800 / 200 / 50 points, ragged against every tile size
                                          ("c2_s0_n12800", 2, 0, 12800)):
        f = synth.make_frame(synth.frame_seed(config, sample), n_points=n_points)
        pyr = opyr.build_pyramid(f['cld'], f['dpt_xyz'], ref_knn_search)
        entry = {"config": config, "sample": sample, "n_points": n_points,
                 "cld_sha256": sha(f['cld']), "dpt_xyz_sha256": sha(f['dpt_xyz'])}
        calls = opyr.knn_calls(pyr, f['dpt_xyz'])
        for k, v in pyr.items():
            entry[k] = {"shape": list(v.shape), "dtype": str(v.dtype), "sha256_raw": sha(v)}
            if k in calls:
                # exact f32 distance ties are ordered by kd-tree traversal in the reference;
                # `sha256` is taken after re-ordering each tie run by index (same neighbours,
                # same distances) -- see oracle.knn.canonical_ties
                canon, ties = oknn.canonical_ties(v, *calls[k])
                k1 = min(v.shape[1] + 1, calls[k][0].shape[0])
                _, d1 = oknn.knn_batch(calls[k][0][None], calls[k][1][None], k1, return_dist=True)
                boundary = int((d1[0, :, v.shape[1] - 1] == d1[0, :, -1]).sum()) if k1 > v.shape[1] else 0
                assert boundary == 0, f"{tag}/{k}: tie across the K-th neighbour, pick another seed"
                entry[k].update({"sha256": sha(canon.astype(v.dtype)), "tie_rows": ties})
            else:
                entry[k]["sha256"] = sha(v)
        hashes[tag] = entry
    with open(os.path.join(HERE, "knn_pyramid_hashes.json"), "w") as fh:
        json.dump(hashes, fh, indent=1, sort_keys=True)
    print("knn goldens written")


def make_ops():
    import torch
    m_ffb6d, m_randla, _ = rh.reference_modules()
    rng = np.random.RandomState(21)
    B, C, M, Np, K, U = 2, 12, 300, 70, 16, 450
    feat = rng.standard_normal((B, C, M)).astype(np.float32)
    pool_idx = rng.randint(0, M, size=(B, Np, K)).astype(np.int64)
    interp_idx = rng.randint(0, M, size=(B, U, 1)).astype(np.int64)
    xyz = rng.rand(B, M, 3).astype(np.float32)
    nei = rng.randint(0, M, size=(B, M, K)).astype(np.int64)
    pc = rng.standard_normal((B, M, 8)).astype(np.float32)
    fs = rng.standard_normal((B, C, 90, K)).astype(np.float32)
    act = (3.0 * rng.standard_normal((B, C, 90, K))).astype(np.float32)
    t = torch.from_numpy
    with torch.no_grad():
        rs = m_ffb6d.FFB6D.random_sample(t(feat).unsqueeze(3), t(pool_idx))
        ni = m_ffb6d.FFB6D.nearest_interpolation(t(feat).unsqueeze(3), t(interp_idx))
        gn = m_randla.Building_block.gather_neighbour(t(pc), t(nei))
        bb = m_randla.Building_block(16)
        rpe = bb.relative_pos_encoding(t(xyz), t(nei))
        scores = torch.softmax(t(act), dim=3)               # RandLANet.py:246
        ap = torch.sum(t(fs) * scores, dim=3, keepdim=True)  # RandLANet.py:247-248
    np.savez_compressed(
        os.path.join(HERE, "ops_small.npz"), feat=feat, pool_idx=pool_idx, interp_idx=interp_idx,
        xyz=xyz, nei=nei, pc=pc, fs=fs, act=act, random_sample=rs.numpy(),
        nearest_interpolation=ni.numpy(), gather_neighbour=gn.numpy(),
        relative_pos_encoding=rpe.numpy(), att_pool=ap.numpy())
    print("ops goldens written")


def make_model_keys():
    model = rh.build_reference_model(n_classes=22, n_pts=12288)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as fh:
        json.dump(keys, fh, indent=0)
    print("state_dict keys written:", len(keys))


def reference_forward(n_classes, frames, pyr, seed=0):
    """Run the unmodified reference FFB6D.forward (ffb6d.py:203-337) on CPU."""
    import torch
    model = rh.build_reference_model(n_classes=n_classes, n_pts=frames['cld'].shape[1])
    model.load_state_dict(synth.synth_state_dict(model, seed))
    model.eval()
    inputs = {
        'rgb': torch.from_numpy(frames['rgb'].astype(np.float32)),
        'cld_rgb_nrm': torch.from_numpy(frames['cld_rgb_nrm']),
        'choose': torch.from_numpy(frames['choose'].astype(np.int64)),
    }
    for k, v in pyr.items():
        inputs[k] = torch.from_numpy(v.astype(np.int64) if v.dtype == np.int32 else v)
    with torch.no_grad():
        end_points = model(inputs)
    return {k: v.numpy() for k, v in end_points.items()}


def make_forward():
    # small frame: 120x160 image, 1024 points (level 3 must keep >= 16 points), 2 frames, 5 classes
    frames = synth.make_batch(7, 2, n_points=1024, height=120, width=160)
    pyr = opyr.build_batch(frames, ref_knn_search)
    ep = reference_forward(5, frames, pyr)
    np.savez_compressed(os.path.join(HERE, "forward_small.npz"), **ep)
    # full size, one frame, strided sample of each output
    frames = synth.make_batch(1, 1, n_points=12288)
    pyr = opyr.build_batch(frames, ref_knn_search)
    ep = reference_forward(22, frames, pyr)
    samp = {k: np.ascontiguousarray(v.reshape(-1)[::97]) for k, v in ep.items()}
    samp.update({k + "/absmax": np.float32(np.abs(v).max()) for k, v in ep.items()})
    np.savez_compressed(os.path.join(HERE, "forward_full_sample.npz"), **samp)
    print("forward goldens written")


if __name__ == "__main__":
    what = sys.argv[1:] or ["knn", "ops", "model", "forward"]
    if "knn" in what:
        make_knn()
    if "ops" in what:
        make_ops()
    if "model" in what:
        make_model_keys()
    if "forward" in what:
        make_forward()
