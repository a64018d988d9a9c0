"""CPU suite: the oracle (our restatement of the reference algorithm) against the golden
vectors produced by the reference itself, and -- where /root/reference exists -- against
the reference's own compiled kd-tree."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import synth
from oracle import knn as oknn
from oracle import ops_ref
from oracle import pyramid as opyr


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def knn_small():
    return np.load(os.path.join(GOLDEN, "knn_small.npz"))


def case_names(z):
    return sorted({k.split("/")[0] for k in z.files})


def test_oracle_knn_matches_reference_goldens(knn_small):
    for name in case_names(knn_small):
        sup, qry = knn_small[name + "/support"], knn_small[name + "/query"]
        K = int(knn_small[name + "/K"])
        got = oknn.knn_batch(sup, qry, K)
        assert got.dtype == np.int64 and got.shape == (sup.shape[0], qry.shape[1], K)
        np.testing.assert_array_equal(got.astype(np.int32), knn_small[name + "/idx"], err_msg=name)


def test_oracle_knn_sorted_and_exact_by_numpy():
    rng = np.random.RandomState(5)
    sup = rng.rand(1, 700, 3).astype(np.float32)
    qry = rng.rand(1, 50, 3).astype(np.float32)
    idx, dist = oknn.knn_batch(sup, qry, 16, return_dist=True)
    assert (np.diff(dist, axis=-1) >= 0).all()
    d = qry[0][:, None, :] - sup[0][None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    ref = np.argsort(d2, axis=1, kind="stable")[:, :16]
    np.testing.assert_array_equal(idx[0], ref)


def test_oracle_knn_duplicate_points_lowest_index_first():
    base = np.random.RandomState(9).rand(40, 3).astype(np.float32)
    sup = np.concatenate([base, base, base], axis=0)[None]  # every point three times
    idx, dist = oknn.knn_batch(sup, base[None], 3, return_dist=True)
    np.testing.assert_array_equal(dist[0], 0.0)
    np.testing.assert_array_equal(idx[0], np.arange(40)[:, None] + np.array([0, 40, 80])[None])


@pytest.mark.parametrize("tag", ["c2_s0_n12288", "c2_s1_n12288"])
def test_oracle_pyramid_matches_reference_hashes(tag):
    with open(os.path.join(GOLDEN, "knn_pyramid_hashes.json")) as fh:
        gold = json.load(fh)[tag]
    f = synth.make_frame(synth.frame_seed(gold["config"], gold["sample"]), n_points=gold["n_points"])
    assert sha(f["cld"]) == gold["cld_sha256"], "synthetic frame generator drifted"
    assert sha(f["dpt_xyz"]) == gold["dpt_xyz_sha256"]
    pyr = opyr.build_pyramid(f["cld"], f["dpt_xyz"], oknn.knn_search)
    calls = opyr.knn_calls(pyr, f["dpt_xyz"])
    for k, v in pyr.items():
        assert list(v.shape) == gold[k]["shape"] and str(v.dtype) == gold[k]["dtype"], k
        if k in calls:   # tie runs in canonical (distance, index) order, see oracle.knn.canonical_ties
            v, _ = oknn.canonical_ties(v, *calls[k])
        assert sha(v.astype(np.int32) if k in calls else v) == gold[k]["sha256"], \
            f"{k} differs from the reference kd-tree result"


def test_ops_ref_matches_reference_goldens():
    z = np.load(os.path.join(GOLDEN, "ops_small.npz"))
    t = torch.from_numpy
    np.testing.assert_array_equal(ops_ref.random_sample(t(z["feat"]), t(z["pool_idx"])).numpy(), z["random_sample"])
    np.testing.assert_array_equal(ops_ref.nearest_interpolation(t(z["feat"]).unsqueeze(3), t(z["interp_idx"])).numpy(),
                                  z["nearest_interpolation"])
    np.testing.assert_array_equal(ops_ref.gather_neighbour(t(z["pc"]), t(z["nei"])).numpy(), z["gather_neighbour"])
    np.testing.assert_allclose(ops_ref.relative_pos_encoding(t(z["xyz"]), t(z["nei"])).numpy(),
                               z["relative_pos_encoding"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(ops_ref.att_pool(t(z["fs"]), t(z["act"])).numpy(), z["att_pool"], rtol=1e-6, atol=1e-6)


def test_oracle_distance_pick_matches_reference_goldens():
    """knn_pick_small.npz = output of the reference's cpp_knn_batch_distance_pick with a pinned clock
    (tests/golden/make_golden_pick.py): the restated std::mt19937, candidate order, counters and K-NN agree."""
    z = np.load(os.path.join(GOLDEN, "knn_pick_small.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        pts, K, seed = z[name + "/pts"], int(z[name + "/K"]), int(z[name + "/seed"])
        idx, q = oknn.knn_batch_distance_pick(pts, z[name + "/idx"].shape[1], K, seed)
        np.testing.assert_array_equal(idx, z[name + "/idx"], err_msg=name)
        np.testing.assert_array_equal(q, z[name + "/queries"], err_msg=name)


# ---- direct comparisons with the reference's own code (build container only) ------------
@pytest.mark.reference
def test_oracle_distance_pick_equals_reference():
    from oracle import ref_harness as rh
    pts = np.random.RandomState(5).rand(2, 700, 3).astype(np.float32)
    for seed in (1, 99991):
        a = oknn.knn_batch_distance_pick(pts, 300, 16, seed)
        b = rh.ref_knn_batch_distance_pick(pts, 300, 16, seed)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.reference
@pytest.mark.parametrize("B,S,Q,K", [(1, 3000, 3000, 16), (2, 777, 1500, 1), (1, 4800, 48, 16), (1, 48, 4800, 1)])
def test_oracle_knn_equals_reference_kdtree(B, S, Q, K):
    from oracle import ref_harness as rh
    rng = np.random.RandomState(S + Q + K)
    sup = rng.rand(B, S, 3).astype(np.float32)
    qry = rng.rand(B, Q, 3).astype(np.float32)
    np.testing.assert_array_equal(oknn.knn_batch(sup, qry, K), rh.ref_knn_batch(sup, qry, K, omp=True))


@pytest.mark.reference
def test_ops_ref_equals_reference_functions():
    from oracle import ref_harness as rh
    m_ffb6d, m_randla, _ = rh.reference_modules()
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(2, 7, 120, 1, generator=g)
    pool = torch.randint(0, 120, (2, 33, 16), generator=g)
    up = torch.randint(0, 120, (2, 500, 1), generator=g)
    xyz = torch.rand(2, 120, 3, generator=g)
    nei = torch.randint(0, 120, (2, 120, 16), generator=g)
    assert torch.equal(ops_ref.random_sample(feat, pool), m_ffb6d.FFB6D.random_sample(feat, pool))
    assert torch.equal(ops_ref.nearest_interpolation(feat, up), m_ffb6d.FFB6D.nearest_interpolation(feat, up))
    pc = torch.randn(2, 120, 9, generator=g)
    assert torch.equal(ops_ref.gather_neighbour(pc, nei), m_randla.Building_block.gather_neighbour(pc, nei))
    bb = m_randla.Building_block(16)
    assert torch.equal(ops_ref.relative_pos_encoding(xyz, nei), bb.relative_pos_encoding(xyz, nei))


@pytest.mark.reference
def test_inputs_ref_equals_the_reference_dpt_2_pcld():
    """oracle/inputs_ref.dpt_2_pcld against the reference's own Dataset.dpt_2_pcld (linemod_dataset.py:188-199, called
    unbound on an object that carries the two index maps its __init__ builds, :31-32) followed by the NaN/Inf clean-up
    of get_item (:258-259): bit-identical, including invalid, NaN and Inf depth pixels and both intrinsics."""
    import types
    from oracle import inputs_ref
    from oracle import ref_harness as rh
    Dataset = rh.reference_dataset_class()
    me = types.SimpleNamespace(xmap=np.array([[j for i in range(640)] for j in range(480)]),
                               ymap=np.array([[i for i in range(640)] for j in range(480)]))
    rng = np.random.RandomState(3)
    dpt = (1000.0 * (0.5 + rng.rand(480, 640))).astype(np.float32)
    dpt[rng.rand(480, 640) < 0.1] = 0.0
    dpt[7, 9], dpt[8, 10] = np.nan, np.inf
    for K, cam_scale in ((synth.LINEMOD_K, 1000.0), (np.array([[1066.778, 0., 312.9869], [0., 1067.487, 241.3109], [0., 0., 1.]]), 10000.0)):
        want = Dataset.dpt_2_pcld(me, dpt.copy(), cam_scale, K)
        want[np.isnan(want)] = 0.0
        want[np.isinf(want)] = 0.0
        got = inputs_ref.dpt_2_pcld(dpt.copy(), cam_scale, K)
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want)


def test_depth_normal_restatement_recovers_plane_normals():
    """normalSpeed is absent here (parity unpinned, oracle/inputs_ref.py): property check of the restated algorithm --
    on a planar depth image z = z0 + gx*x + gy*y (mm per pixel) the LINE-MOD normal is normalize(fx*gx, fy*gy, -z);
    the r-wide border, zero-depth regions and pixels at or beyond the distance threshold stay (0, 0, 0)."""
    from oracle import inputs_ref
    yy, xx = np.mgrid[0:100, 0:140]
    fx, fy = 572.4, 573.6
    for gx, gy in ((2.0, 1.0), (-1.0, 0.5), (0.0, 0.0)):
        z = 1000.0 + gx * xx + gy * yy
        n = inputs_ref.depth_normal(z.astype(np.float32), fx, fy, 5, 2000, 20, False)
        want = np.stack([fx * gx * np.ones_like(z), fy * gy * np.ones_like(z), -np.floor(z)], axis=-1)
        want /= np.linalg.norm(want, axis=-1, keepdims=True)
        np.testing.assert_allclose(n[5:94, 5:134], want[5:94, 5:134], atol=2e-3)
        assert (n[:5] == 0).all() and (n[:, :5] == 0).all() and (n[94:] == 0).all() and (n[:, 134:] == 0).all()
    z = np.full((60, 60), 2000.0, np.float32)
    assert (inputs_ref.depth_normal(z, fx, fy) == 0).all()                       # d < distance_threshold is strict
    z[:] = 0.0
    assert (inputs_ref.depth_normal(z, fx, fy) == 0).all()                       # invalid depth: degenerate system


# ---- whole-forward restatement (oracle/forward_ref.py) against the reference's end_points ----
def _oracle_forward(config, bs, n_points, h, w, n_classes):
    from oracle import forward_ref
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        shapes = json.load(fh)
    sd = synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=n_classes)
    frames = synth.make_batch(config, bs, n_points=n_points, height=h, width=w)
    pyr = opyr.build_batch(frames, oknn.knn_search)
    inputs = {"rgb": torch.from_numpy(frames["rgb"].astype(np.float32)),
              "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]),
              "choose": torch.from_numpy(frames["choose"].astype(np.int64))}
    for k, v in pyr.items():
        inputs[k] = torch.from_numpy(v.astype(np.int64) if v.dtype == np.int32 else v)
    with torch.no_grad():
        return forward_ref.ffb6d_forward(sd, inputs)


def test_oracle_forward_matches_reference_small_golden():
    """2 frames of 120x160, N=1024, 5 classes: full end_points of the reference FFB6D."""
    gold = np.load(os.path.join(GOLDEN, "forward_small.npz"))
    ep = _oracle_forward(7, 2, 1024, 120, 160, 5)
    for k in ("pred_rgbd_segs", "pred_kp_ofs", "pred_ctr_ofs"):
        assert ep[k].shape == gold[k].shape
        scale = float(np.abs(gold[k]).max())
        err = float(np.abs(ep[k].numpy() - gold[k]).max())
        assert err <= 1e-5 * max(scale, 1.0), (k, err, scale)


def test_depth_backprojection_restatement_matches_the_frame_generator():
    """oracle/inputs_ref.dpt_2_pcld (line-by-line restatement of linemod_dataset.py:188-199; the
    dataset module itself cannot be imported here: cv2/normalSpeed are absent) must agree with the
    back-projection inside ffb6d_amd.synth.make_frame, which every golden frame went through."""
    from oracle import inputs_ref
    f = synth.make_frame(2000, n_points=1024, height=120, width=160)
    want = inputs_ref.dpt_2_pcld(f["dpt_xyz"][2], 1.0, synth.LINEMOD_K).astype(np.float32).transpose(2, 0, 1)
    np.testing.assert_array_equal(want, f["dpt_xyz"])
    z = f["dpt_xyz"][2].copy()
    z[0, 0] = np.nan
    assert (inputs_ref.dpt_2_pcld(z, 1.0, synth.LINEMOD_K)[0, 0] == 0).all()


def test_fill_missing_restatement_fills_holes_and_keeps_the_surface():
    """oracle/holefill_ref.py is unpinned (no cv2 here); these are the properties IP-Basic's completion guarantees: holes
    below each column's first valid pixel are filled, nothing is invented more than the unconditional dilations' reach
    (3 + 2 rows) above it, and the filled surface stays within a few centimetres of a smooth ground truth."""
    from oracle import holefill_ref
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:96, 0:128]
    truth = 0.7 + 0.01 * xx + 0.003 * yy
    d = (truth * 10000.0).astype(np.uint16)
    d[rng.rand(96, 128) < 0.3] = 0
    d[:10] = 0
    out = holefill_ref.fill_missing(d, 10000.0, 1)
    assert out.dtype == np.float32 and out.shape == d.shape
    assert (out[:5] == 0).all()
    assert (out[16:] > 0).all()
    assert np.abs(out[16:] / 10000.0 - truth[16:]).max() < 0.05
    # the morphology helpers have OpenCV's border rule: out-of-image pixels never win
    img = np.full((6, 6), -2.0, np.float32)
    assert (holefill_ref.dilate(img, holefill_ref.full(5)) == -2.0).all() and (holefill_ref.erode(img, holefill_ref.full(5)) == -2.0).all()
    assert holefill_ref.CROSS_3.sum() == 5 and holefill_ref.CROSS_5.sum() == 9 and holefill_ref.CROSS_7.sum() == 13
