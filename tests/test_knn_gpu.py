"""GPU parity: the HIP KNN (through the C ABI) against the oracle and the reference goldens.
Bar: indices bit-exact (BASELINE.json north_star)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import nearest_neighbors as nn
from ffb6d_amd import pyramid, synth
from ffb6d_amd.helper_tool import DataProcessing as DP
from oracle import knn as oknn
from oracle import pyramid as opyr

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_device_knn_matches_reference_goldens(device):
    z = np.load(os.path.join(GOLDEN, "knn_small.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        sup = torch.from_numpy(z[name + "/support"]).to(device)
        qry = torch.from_numpy(z[name + "/query"]).to(device)
        K = int(z[name + "/K"])
        for dt in (torch.int64, torch.int32):
            got = nn.knn_batch_device(sup, qry, K, dtype=dt)
            assert got.dtype == dt
            np.testing.assert_array_equal(got.cpu().numpy().astype(np.int32), z[name + "/idx"], err_msg=name)


@pytest.mark.parametrize("B,S,Q,K", [
    (1, 12288, 12288, 16),   # self-KNN, level 0
    (2, 3072, 12288, 1),     # up-sampling 1-NN
    (1, 19200, 3072, 16),    # r2p: image grid support
    (2, 76800, 768, 16),     # split-S path (few queries, big grid)
    (1, 768, 76800, 1),      # p2r: many queries
    (3, 48, 192, 1), (3, 192, 192, 16), (1, 16, 5, 16), (1, 1000, 1, 16), (1, 1023, 257, 7),
    (1, 2049, 255, 32), (4, 4800, 48, 16), (1, 5000, 300, 2),
])
def test_device_knn_matches_oracle(device, B, S, Q, K):
    rng = np.random.RandomState(B * 7 + S + Q + K)
    sup = rng.rand(B, S, 3).astype(np.float32)
    qry = sup.copy() if S == Q else rng.rand(B, Q, 3).astype(np.float32)
    want_i, want_d = oknn.knn_batch(sup, qry, K, return_dist=True)
    got_i, got_d = nn.knn_batch_device(torch.from_numpy(sup).to(device), torch.from_numpy(qry).to(device), K,
                                       return_dist=True)
    np.testing.assert_array_equal(got_i.cpu().numpy(), want_i)
    np.testing.assert_array_equal(got_d.cpu().numpy(), want_d)   # distances are bit-exact too


def test_host_pointer_entry_points_match_oracle(device):
    """cpp_knn / cpp_knn_omp / cpp_knn_batch / cpp_knn_batch_omp (names of knn_.h:4-19)."""
    rng = np.random.RandomState(1)
    sup = rng.rand(2, 900, 3).astype(np.float32)
    qry = rng.rand(2, 400, 3).astype(np.float32)
    want = oknn.knn_batch(sup, qry, 16)
    for omp in (False, True):
        got = nn.knn_batch(sup, qry, 16, omp=omp)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(nn.knn(sup[0], qry[0], 16, omp=omp), want[0])
    got = DP.knn_search(sup, qry, 16)
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, want.astype(np.int32))
    # float64 / non-contiguous input is converted like knn.pyx:95-96
    np.testing.assert_array_equal(nn.knn_batch(sup.astype(np.float64)[:, ::-1][:, ::-1], qry, 16), want)


def test_duplicate_points_compare_by_distance(device):
    """np.pad(..., 'wrap')-style clouds (linemod_dataset.py:276-277) contain exact
    duplicates: the kd-tree's pick among equal distances is traversal dependent, ours is the
    lowest index; the distance rows must agree exactly with the oracle either way."""
    base = np.random.RandomState(4).rand(700, 3).astype(np.float32)
    sup = np.concatenate([base, base[:324]], axis=0)[None]
    qry = sup.copy()
    want_i, want_d = oknn.knn_batch(sup, qry, 16, return_dist=True)
    got_i, got_d = nn.knn_batch_device(torch.from_numpy(sup).to(device), torch.from_numpy(qry).to(device), 16,
                                       return_dist=True)
    np.testing.assert_array_equal(got_d.cpu().numpy(), want_d)
    np.testing.assert_array_equal(got_i.cpu().numpy(), want_i)   # same tie rule as the oracle
    gi = got_i.cpu().numpy()[0]
    assert (gi[:, 0] == np.where(np.arange(1024) >= 700, np.arange(1024) - 700, np.arange(1024))).all()


def test_invalid_pixels_at_origin(device):
    """Invalid-depth pixels are all (0,0,0) in the image grids (linemod_dataset.py:198)."""
    f = synth.make_frame(31, n_points=768, height=120, width=160)
    grid = synth.strided_grids(f["dpt_xyz"])[4][None]
    assert (np.abs(grid).sum(-1) == 0).sum() > 10
    cld = f["cld"][None]
    for sup, qry, K in ((grid, cld, 16), (cld, grid, 1)):
        want = oknn.knn_batch(sup, qry, K)
        got = nn.knn_batch_device(torch.from_numpy(sup).to(device), torch.from_numpy(qry).to(device), K)
        # queries AT the origin are tied between all zero pixels only when the support has them
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("tag", ["c2_s0_n12288", "c2_s1_n12288", "c4_s0_n24576", "c2_s0_n12800"])
def test_full_size_pyramid_matches_reference_hashes(device, tag):
    """All 26 index tensors of a 480x640 frame, bit-exact against what the reference's own
    nanoflann produced (hashes committed by tests/golden/make_golden.py)."""
    with open(os.path.join(GOLDEN, "knn_pyramid_hashes.json")) as fh:
        gold = json.load(fh)[tag]
    f = synth.make_frame(synth.frame_seed(gold["config"], gold["sample"]), n_points=gold["n_points"])
    assert sha(f["cld"]) == gold["cld_sha256"]
    cld = torch.from_numpy(f["cld"][None]).to(device)
    dpt = torch.from_numpy(f["dpt_xyz"][None]).to(device)
    pyr = pyramid.build_index_pyramid(cld, dpt, index_dtype=torch.int32)
    host = {k: v[0].cpu().numpy() for k, v in pyr.items()}
    calls = opyr.knn_calls(host, f["dpt_xyz"])
    for k, a in host.items():
        assert list(a.shape) == gold[k]["shape"], k
        if k in calls:
            # exact-distance ties: the kd-tree orders them by traversal, we by index; both
            # are mapped to the canonical (distance, index) order before hashing
            canon, _ = oknn.canonical_ties(a, *calls[k])
            assert np.array_equal(canon, a), f"{k}: our tie order must already be canonical"
        assert sha(a) == gold[k]["sha256"], f"{k} differs from the reference kd-tree result"
        if k in calls and gold[k]["tie_rows"] == 0:
            assert sha(a) == gold[k]["sha256_raw"]
    pyr64 = pyramid.build_index_pyramid(cld, dpt, index_dtype=torch.int64)
    for k, v in pyr64.items():
        assert torch.equal(v.to(pyr[k].dtype), pyr[k]), k


def test_batched_pyramid_equals_per_frame(device):
    frames = synth.make_batch(2, 3, n_points=1024, height=120, width=160)
    cld = torch.from_numpy(frames["cld"]).to(device)
    dpt = torch.from_numpy(frames["dpt_xyz"]).to(device)
    batched = pyramid.build_index_pyramid(cld, dpt)
    for b in range(3):
        single = pyramid.build_index_pyramid(cld[b:b + 1], dpt[b:b + 1])
        for k in single:
            assert torch.equal(batched[k][b:b + 1], single[k]), (k, b)


def test_knn_rejects_bad_arguments(device):
    from ffb6d_amd import _lib
    s = torch.zeros(1, 10, 3, device=device)
    with pytest.raises(ValueError):
        nn.knn_batch_device(s, s, 16)
    with pytest.raises(TypeError):
        nn.knn_batch_device(s.double(), s.double(), 4)
    with pytest.raises(_lib.FFB6DNativeError):
        nn.knn_batch_device(s.cpu(), s.cpu(), 4)
    e = nn.knn_batch_device(s, torch.zeros(1, 0, 3, device=device), 4)   # empty query set
    assert e.shape == (1, 0, 4)


def test_prepared_sets_and_raw_queries_agree_with_the_oracle(device):
    """ffb6d_knn_prepare / ffb6d_knn_search_prepared: a prepared support serves prepared and raw
    (unsorted) query sets; results equal the brute-force oracle bit for bit."""
    rng = np.random.RandomState(77)
    sup = rng.rand(2, 5000, 3).astype(np.float32)
    qry = rng.rand(2, 333, 3).astype(np.float32)
    want_i, want_d = oknn.knn_batch(sup, qry, 16, return_dist=True)
    ps = nn.PreparedPoints(torch.from_numpy(sup).to(device))
    q_dev = torch.from_numpy(qry).to(device)
    got_i, got_d = nn.knn_prepared(ps, q_dev, 16, return_dist=True)                      # raw queries
    np.testing.assert_array_equal(got_i.cpu().numpy(), want_i)
    np.testing.assert_array_equal(got_d.cpu().numpy(), want_d)
    got_i2 = nn.knn_prepared(ps, nn.PreparedPoints(q_dev), 16, dtype=torch.int32)         # prepared queries
    np.testing.assert_array_equal(got_i2.cpu().numpy(), want_i.astype(np.int32))
    got_1 = nn.knn_prepared(ps, nn.PreparedPoints(q_dev), 1)                              # K = 1 kernel
    np.testing.assert_array_equal(got_1.cpu().numpy(), oknn.knn_batch(sup, qry, 1))
    with pytest.raises(ValueError):
        nn.knn_prepared(ps, q_dev, 1)                                                     # raw needs 2 <= K <= 16


def test_sets_prepared_together_equal_sets_prepared_one_by_one(device):
    """ffb6d_knn_prepare_multi (one Morton sort over the concatenation of all sets, the set as one more grid dimension of
    every other pass) writes byte-identical prepared sets to ffb6d_knn_prepare -- what the index-pyramid builder relies on"""
    rng = np.random.RandomState(5)
    sets = [torch.from_numpy(rng.rand(3, n, 3).astype(np.float32) * s).to(device) for n, s in ((5000, 1.0), (64, 3.0), (2049, 0.5), (777, 1.0))]
    sets.append(sets[0][:, :1250].contiguous())                     # a prefix level, as in the pyramid
    many = nn.prepare_many(sets)
    for p, m in zip(sets, many):
        one = nn.PreparedPoints(p)
        assert m.S == one.S and torch.equal(m.blob[:-256], one.blob[:-256])     # (the last < 256 bytes are alignment padding)
    assert len(nn.prepare_many(sets * 2)) == 10                     # more than 8 sets: chunked


def test_distance_pick_matches_reference_goldens_and_oracle(device, monkeypatch):
    """cpp_knn_batch_distance_pick[_omp] (knn_.h:21-27) through the C ABI: bit-exact against the reference's own
    output (knn_pick_small.npz, clock pinned) and against the oracle on a bigger frame."""
    z = np.load(os.path.join(GOLDEN, "knn_pick_small.npz"))
    for name in sorted({k.split("/")[0] for k in z.files}):
        pts, K, seed = z[name + "/pts"], int(z[name + "/K"]), int(z[name + "/seed"])
        monkeypatch.setenv("FFB6D_KNN_PICK_SEED", str(seed))
        for omp in (False, True):
            idx, q = nn.knn_batch_distance_pick(pts, z[name + "/idx"].shape[1], K, omp=omp)
            np.testing.assert_array_equal(idx, z[name + "/idx"], err_msg=name)
            np.testing.assert_array_equal(q, z[name + "/queries"], err_msg=name)
    pts = np.random.RandomState(11).rand(2, 3072, 3).astype(np.float32)
    monkeypatch.setenv("FFB6D_KNN_PICK_SEED", "4242")
    idx, q = nn.knn_batch_distance_pick(pts, 1000, 16)
    want_i, want_q = oknn.knn_batch_distance_pick(pts, 1000, 16, 4242)
    np.testing.assert_array_equal(idx, want_i)
    np.testing.assert_array_equal(q, want_q)
    # every query is one of the frame's points and its own nearest neighbour
    assert (idx[..., 0] == np.array([[np.flatnonzero((pts[b] == q[b, i]).all(1))[0] for i in range(1000)]
                                     for b in range(2)])).all()
