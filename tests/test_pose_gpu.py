"""GPU parity of the pose solver (ffb6d_amd/pose.py over include/ffb6d_pose.h) against
  * outputs of the reference itself (tests/golden/pose_small.npz), and
  * the oracle restatement (oracle/pose_ref.py) on further seeded cases.

Tolerances.  Mean shift is an fp32 fixed-point iteration; the kernel sums the Gaussian-weighted
mean in another order than torch's CPU reduction and uses the hardware exp2, so converged points
differ by rounding noise and a set may stop one round earlier or later.  The loop stops when no
point moved more than stop_thresh = bandwidth*1e-3 (4e-5 m) in a round -- the points of a cluster
are then still a few stop_thresh apart -- and the answer is ONE of them, picked by arg-max ball
size, where neighbouring points differ by a handful of counts: rounding may pick another point of
the same cluster.  Hence:
    centres / keypoints   |diff| <= 5e-4 m   (12 stop_thresh; the synthetic vote noise is 4e-3 m)
    labels                <= 0.5 % of a set's points may flip (points on the ball's surface)
    poses [R|t]           <= 5e-3 (9-point Kabsch of keypoints that each carry <= 5e-4 m over a
                          0.2 m model)
best_fit_transform alone (same keypoints in): the reference runs LAPACK in float32, the kernel
Jacobi in float64 -> |diff| <= 2e-5.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import pose, synth

pytestmark = pytest.mark.gpu

spec = importlib.util.spec_from_file_location("make_golden_pose", os.path.join(GOLDEN, "make_golden_pose.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

CTR_TOL = 5e-4
LABEL_FLIP = 0.005
POSE_TOL = 5e-3
BFT_TOL = 2e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "pose_small.npz"))


def dev_case(case, device):
    return (torch.from_numpy(case["pcld"]).to(device), torch.from_numpy(case["mask"]).to(device),
            torch.from_numpy(case["ctr_of"]).to(device), torch.from_numpy(case["kp_of"]).to(device))


@pytest.mark.parametrize("i", range(len(gen.MS_CASES)))
def test_mean_shift_matches_reference(device, gold, i):
    seed, n, bw = gen.MS_CASES[i]
    votes = torch.from_numpy(gen.ms_votes(seed, n)).to(device)
    ctr, lab = pose.MeanShiftTorch(bandwidth=bw).fit(votes)
    assert lab.dtype == torch.bool and lab.shape == (n,)
    assert np.abs(ctr.cpu().numpy() - gold[f"ms{i}_ctr"]).max() <= CTR_TOL
    flips = int((lab.cpu().numpy() != gold[f"ms{i}_labels"]).sum())
    assert flips <= max(1, int(LABEL_FLIP * n)), flips


def test_mean_shift_batch_equals_single_fits(device):
    """Sets of different sizes in one launch: each must equal its own single-set run bit for bit
    (no cross-talk through padding, shared tiles or the per-set stopping flags)."""
    clouds = [torch.from_numpy(gen.ms_votes(500 + k, n)).to(device) for k, n in enumerate((40, 513, 1, 2000, 64, 700))]
    ms = pose.MeanShiftTorch(bandwidth=0.04)
    centers, labels = ms.fit_batch(clouds)
    for g, c in enumerate(clouds):
        ctr, lab = ms.fit(c)
        assert torch.equal(ctr, centers[g]) and torch.equal(lab, labels[g])


def test_mean_shift_sets_beyond_the_one_workgroup_limit(device):
    """Round 5: sets of up to 4096 points are fitted by one workgroup each (duplicate merging, csrc/pose.hip mean_shift_fit_kernel), larger
    ones by the round-by-round kernels -- both in one call here, each against the oracle's restatement of MeanShiftTorch.fit."""
    from oracle import pose_ref
    sizes = (4200, 300, 4096)
    clouds = [gen.ms_votes(520 + k, n) for k, n in enumerate(sizes)]
    centers, labels = pose.MeanShiftTorch(bandwidth=0.05).fit_batch([torch.from_numpy(c).to(device) for c in clouds])
    for g, c in enumerate(clouds):
        want_c, want_l, _ = pose_ref.mean_shift_fit(torch.from_numpy(c), 0.05)
        assert np.abs(centers[g].cpu().numpy() - np.asarray(want_c)).max() <= CTR_TOL, sizes[g]
        flips = int((labels[g].cpu().numpy() != np.asarray(want_l)).sum())
        assert flips <= max(1, int(LABEL_FLIP * sizes[g])), (sizes[g], flips)


def test_mean_shift_forms_give_the_same_bits(device):
    """Round 6: the first two rounds of the sets of 512 .. 4096 points are made chip-wide (mean_shift_spread_round_kernel) and the
    one-workgroup fits start from them; sets of up to 2048 points run on the light fit.  Same pair arithmetic in every form: the
    centres, labels, ball sizes and round counts must be equal bit for bit -- sizes either side of every threshold, sets whose
    first round already merges duplicates (round 1's spread results are then not used), ragged counts behind one stride."""
    from ffb6d_amd import _lib
    lib = _lib.load()
    sizes = (511, 512, 513, 1700, 2048, 2049, 3000, 4096, 40, 0, 900, 640)
    stride = 4096
    sets = torch.zeros((len(sizes), stride, 4), device=device)
    for g, n in enumerate(sizes):
        if n:
            v = torch.from_numpy(gen.ms_votes(700 + g, n)).to(device)
            if g in (10, 11):                                  # every vote of these sets twice: round 0 finds duplicates
                v[n // 2:] = v[:n - n // 2]
            sets[g, :n, :3] = v
    counts = torch.tensor(sizes, dtype=torch.int32, device=device)
    got = {}
    try:
        for form, spread in ((1, 2), (1, 1), (1, 0), (0, 2), (0, 0)):
            lib.ffb6d_pose_set_fit_form(form)
            lib.ffb6d_pose_set_fit_spread(spread)
            got[form, spread] = [t.clone() for t in pose.mean_shift(sets, counts, 0.04)]
            for limit in (0, 1, 2):                            # round limits inside the spread rounds
                got[form, spread] += [t.clone() for t in pose.mean_shift(sets, counts, 0.04, limit)]
    finally:
        lib.ffb6d_pose_set_fit_form(1)
        lib.ffb6d_pose_set_fit_spread(1)
    base = got[0, 0]
    assert int(base[3][3]) > 3 and int(base[3][10]) > 3        # real fits, not early exits
    for key, res in got.items():
        for a, b in zip(base, res):
            assert torch.equal(a, b), key


def test_mean_shift_large_sets_chip_wide_rounds_then_one_workgroup(device):
    """Round 6: sets of more than 4096 points run their rounds chip-wide on (position, multiplicity) lists with exact-duplicate merging
    and continue in the one-workgroup fit once at most 4096 distinct positions are left (csrc/pose.hip big_round_kernel /
    big_compact_kernel); before, they made max_iter + 1 rounds of count^2 pairs.  Both forms in one process, on clustered votes
    (collapse within ~10 rounds), a scene-like cloud (collapses later), votes that come in pairs of equal points, and a lattice whose
    points never meet (stays with the round-by-round kernels: equal bits); mixed with small sets in the same call; round limits that
    end inside the chip-wide phase.  The 4200-point case is also held against the oracle in the test above."""
    from ffb6d_amd import _lib
    lib = _lib.load()
    kBigPhase = 8                                               # rounds before anything of a scene-like cloud has merged
    rng = np.random.RandomState(12)
    sizes = (9000, 300, 12288, 4097, 6000, 5832, 2500, 7000)
    clouds = [gen.ms_votes(800 + k, n) for k, n in enumerate(sizes)]
    clouds[2] = (rng.rand(12288, 3).astype(np.float32) - 0.5) * np.float32([1.2, 1.0, 0.5])          # scene-like: many modes
    clouds[4][3000:] = clouds[4][:3000]                                                              # every vote twice
    g = np.arange(18, dtype=np.float32) * 0.2
    clouds[5] = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)                     # 18^3 points 5 bandwidths apart
    stride = 12288
    sets = torch.zeros((len(sizes), stride, 4), device=device)
    for k, c in enumerate(clouds):
        sets[k, :len(c), :3] = torch.from_numpy(c).to(device)
    counts = torch.tensor(sizes, dtype=torch.int32, device=device)
    res = {}
    try:
        for form in (0, 1):
            lib.ffb6d_pose_set_big_form(form)
            for limit in (300, 0, 1, 3, 7):
                res[form, limit] = [t.clone() for t in pose.mean_shift(sets, counts, 0.04, limit)]
    finally:
        lib.ffb6d_pose_set_big_form(1)
    for limit in (300, 0, 1, 3, 7):
        (c0, l0, n0, r0), (c1, l1, n1, r1) = res[0, limit], res[1, limit]
        assert (r0 - r1).abs().max() <= (1 if limit == 300 else 0), (limit, r0.tolist(), r1.tolist())
        for k, n in enumerate(sizes):
            if n <= 4096 or k == 5:                             # the same kernels in both forms
                assert torch.equal(c0[k], c1[k]) and torch.equal(l0[k], l1[k]) and int(n0[k]) == int(n1[k]), (limit, k)
                continue
            if limit == 300 and k == 2:
                # many modes of nearly equal ball size: the arg-max may land in another mode; the winner's ball must be as large
                assert abs(int(n0[k]) - int(n1[k])) <= max(2, int(0.01 * int(n0[k]))), (int(n0[k]), int(n1[k]))
                continue
            assert abs(int(n0[k]) - int(n1[k])) <= max(1, int(LABEL_FLIP * n)), (limit, k)
            if limit != 300:
                # stopped before the collapse: the arg-max picks among thousands of points whose ball sizes differ by a handful --
                # rounding may pick a neighbour (inside the same ball)
                assert (c0[k] - c1[k]).abs().max() <= 0.04, (limit, k, c0[k].tolist(), c1[k].tolist())
                continue
            assert (c0[k] - c1[k]).abs().max() <= CTR_TOL, (limit, k, c0[k].tolist(), c1[k].tolist())
            flips = int((l0[k] != l1[k]).sum())
            assert flips <= max(1, int(LABEL_FLIP * n)), (limit, k, flips)
    assert int(res[1, 300][3][2]) > kBigPhase, res[1, 300][3].tolist()        # the scene-like set went on in the one-workgroup fit


def test_mean_shift_round_limit_and_polling(device):
    """max_iter bounds the rounds (it > max_iter after max_iter+1 rounds, meanshift_pytorch.py:47);
    polling the stop flag every k rounds must not change the result."""
    votes = torch.from_numpy(gen.ms_votes(600, 1200)).to(device)
    sets = torch.zeros((1, 1200, 4), device=device)
    sets[0, :, :3] = votes
    counts = torch.tensor([1200], dtype=torch.int32, device=device)
    c0, l0, n0, r0 = pose.mean_shift(sets, counts, 0.04, 300, check_every=0)
    c1, l1, n1, r1 = pose.mean_shift(sets, counts, 0.04, 300, check_every=1)
    c8, l8, n8, r8 = pose.mean_shift(sets, counts, 0.04, 300, check_every=8)
    assert torch.equal(c0, c1) and torch.equal(c0, c8) and torch.equal(l0, l8) and int(r0) == int(r1) == int(r8)
    assert 1 <= int(r0) <= 301 and int(n0) == int(l0.sum())
    _, _, _, r = pose.mean_shift(sets, counts, 0.04, 2, check_every=0)
    assert int(r) == 3


@pytest.mark.parametrize("i", range(6))
def test_best_fit_transform_matches_reference(device, gold, i):
    A, B = gen.bft_case(400 + i)
    T = pose.best_fit_transform(A, B)
    assert T.shape == (3, 4) and T.dtype == np.float64
    assert np.abs(T - gold[f"bft{i}_T"]).max() <= BFT_TOL
    assert abs(np.linalg.det(T[:, :3]) - 1) < 1e-9


def test_best_fit_transform_degenerate_inputs(device):
    rng = np.random.RandomState(0)
    A = rng.rand(9, 3).astype(np.float32)
    T = pose.best_fit_transform(A, A)                                  # identity
    assert np.abs(T - np.eye(4)[:3]).max() < 1e-6
    line = np.outer(np.linspace(-1, 1, 9), [1, 2, 3]).astype(np.float32)   # rank-1 model
    T = pose.best_fit_transform(line, line + np.float32(0.5))
    assert abs(np.linalg.det(T[:, :3]) - 1) < 1e-9
    assert np.abs(line @ T[:, :3].T + T[:, 3] - (line + 0.5)).max() < 1e-5
    T = pose.best_fit_transform(np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32))   # H = 0
    assert np.abs(T[:, :3] - np.eye(3)).max() == 0 and np.abs(T[:, 3] - 1).max() < 1e-12


@pytest.mark.parametrize("i", range(len(gen.LM_CASES)))
def test_linemod_flow_matches_reference(device, gold, i):
    seed, n, n_obj, flt = gen.LM_CASES[i]
    case = synth.make_pose_case(seed, n_pts=n, n_obj=n_obj)
    poses = pose.cal_frame_poses_lm(*dev_case(case, device), True, n_obj + 1, flt, 1,
                                    mesh_kps=case["mesh_kps"][1], mesh_ctr=case["mesh_ctr"][1])
    assert len(poses) == 1
    err = np.abs(poses[0] - gold[f"lm{i}_pose"][0]).max()
    print("lm", i, "pose err", err)
    assert err <= POSE_TOL


@pytest.mark.parametrize("i", range(len(gen.YCB_CASES)))
def test_ycb_flow_matches_reference(device, gold, i):
    seed, n, n_obj, flt = gen.YCB_CASES[i]
    case = synth.make_pose_case(seed, n_pts=n, n_obj=n_obj)
    ids, poses, kps = pose.cal_frame_poses(*dev_case(case, device), True, n_obj + 1, flt,
                                           mesh_kps=case["mesh_kps"], mesh_ctr=case["mesh_ctr"], r_lst=case["r_lst"])
    assert np.array_equal(ids, gold[f"ycb{i}_ids"])
    kerr = np.abs(np.stack(kps) - gold[f"ycb{i}_kps"]).max()
    perr = np.abs(np.stack(poses) - gold[f"ycb{i}_pose"]).max()
    print("ycb", i, "kps err", kerr, "pose err", perr)
    assert kerr <= CTR_TOL and perr <= POSE_TOL


def test_batched_solver_equals_per_frame_and_oracle(device):
    """Frames of different content in one batch (int32 mask, one frame without foreground, one class
    that loses all its points) against per-frame oracle runs."""
    from oracle import pose_ref
    cases = [synth.make_pose_case(700 + k, n_pts=1500, n_obj=3, n_cls=5, mesh_seed=7) for k in range(3)]
    cases[1]["mask"][:] = 0                                              # nothing to solve in frame 1
    cases[2]["mask"][cases[2]["mask"] == 2] = 0                          # class 2 absent in frame 2
    stack = lambda key: torch.from_numpy(np.stack([c[key] for c in cases])).to(device)
    mesh_kps, mesh_ctr, r_lst = cases[0]["mesh_kps"], cases[0]["mesh_ctr"], cases[0]["r_lst"]
    res = pose.solve_poses(stack("pcld"), stack("mask").int(), stack("ctr_of"), stack("kp_of"),
                           mesh_kps, mesh_ctr, r_lst=r_lst)
    assert len(res) == 3 and len(res[1][0]) == 0
    for b, case in enumerate(cases):
        ids, poses, kps = pose_ref.frame_poses_ycb(*gen.tensors(case), True, True, mesh_kps, mesh_ctr, r_lst)
        assert np.array_equal(res[b][0], ids)
        if len(ids):
            assert np.abs(res[b][2] - np.stack(kps)).max() <= CTR_TOL
            assert np.abs(res[b][1] - np.stack(poses)).max() <= POSE_TOL
            for c, T in zip(ids, res[b][1]):
                if (case["mask"] == c).sum() > 50:
                    assert np.abs(T - case["RT"][c]).max() < 0.03


def test_full_size_cloud_recovers_the_pose(device):
    """N = 12288 points, 5 objects: size-independent property -- the solver returns the pose the
    votes were synthesised from."""
    case = synth.make_pose_case(800, n_pts=12288, n_obj=5)
    ids, poses, _ = pose.cal_frame_poses(*dev_case(case, device), True, 6, True, mesh_kps=case["mesh_kps"],
                                         mesh_ctr=case["mesh_ctr"], r_lst=case["r_lst"])
    assert list(ids) == [1, 2, 3, 4, 5]
    for c, T in zip(ids, poses):
        assert np.abs(T - case["RT"][c]).max() < 0.01


def test_mean_shift_edge_sets(device):
    """Empty set, one point, all votes identical, two points further apart than the bandwidth."""
    ms = pose.MeanShiftTorch(bandwidth=0.04)
    one = torch.tensor([[0.1, 0.2, 0.3]], device=device)
    ctr, lab = ms.fit(one)
    assert torch.equal(ctr, one[0]) and lab.tolist() == [True]
    same = one.repeat(50, 1)
    ctr, lab = ms.fit(same)
    assert torch.allclose(ctr, one[0], atol=1e-6) and bool(lab.all())
    far = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.5, 1.0], [0.0, 0.5, 1.0]], device=device)
    ctr, lab = ms.fit(far)                                   # the doubled point wins, the lone one is outside
    assert torch.allclose(ctr, far[1], atol=1e-6) and lab.tolist() == [False, True, True]
    sets = torch.zeros((3, 8, 4), device=device)
    sets[1, :3, :3] = far
    counts = torch.tensor([0, 3, 0], dtype=torch.int32, device=device)
    centers, labels, n_in, rounds = pose.mean_shift(sets, counts, 0.04)
    assert centers[0].abs().max() == 0 and centers[2].abs().max() == 0 and n_in.tolist() == [0, 2, 0]
    assert not bool(labels[0].any()) and labels[1].tolist()[:3] == [False, True, True] and int(rounds[0]) == 0
    with pytest.raises(ValueError):
        ms.fit(torch.zeros((0, 3), device=device))


def test_linemod_flow_without_object_points_returns_identity(device):
    case = synth.make_pose_case(55, n_pts=500, n_obj=1)
    case["mask"][:] = 0
    poses = pose.cal_frame_poses_lm(*dev_case(case, device), True, 2, False, 1,
                                    mesh_kps=case["mesh_kps"][1], mesh_ctr=case["mesh_ctr"][1])
    assert np.array_equal(poses[0], np.identity(4)[:3])     # pvn3d_eval_utils_kpls.py:239-240


def test_cpu_tensors_are_rejected():
    from ffb6d_amd import _lib
    with pytest.raises(_lib.FFB6DNativeError):
        pose.MeanShiftTorch(0.04).fit(torch.zeros(10, 3))
