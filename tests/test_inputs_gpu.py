"""GPU parity of the on-device input pipeline against the numpy
restatement of the dataset code (oracle/inputs_ref.py)."""
import numpy as np
import pytest
import torch

from ffb6d_amd import inputs, synth
from oracle import inputs_ref

pytestmark = pytest.mark.gpu


def _depth(seed, h=120, w=160):
    rng = np.random.RandomState(seed)
    d = (1000.0 * (1.0 + 0.3 * rng.rand(h, w))).astype(np.float32)     # millimetres
    d[rng.rand(h, w) < 0.07] = 0.0
    d[3, 5] = np.nan
    d[4, 6] = np.inf
    return d


def test_depth_to_cloud_is_bit_exact(device):
    K = synth.LINEMOD_K
    deps = np.stack([_depth(1), _depth(2)])
    want = np.stack([inputs_ref.dpt_2_pcld(d, 1000.0, K).astype(np.float32).transpose(2, 0, 1) for d in deps])
    got = inputs.depth_to_cloud(torch.from_numpy(deps).to(device), K, cam_scale=1000.0).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert (got[:, :, 3, 5] == 0).all() and (got[:, :, 4, 6] == 0).all()


def test_sample_choose_properties(device):
    d = torch.from_numpy(np.stack([_depth(3), _depth(4)])).to(device)
    g = torch.Generator(device=device).manual_seed(5)
    ch = inputs.sample_choose(d, 2048, generator=g)
    assert ch.shape == (2, 1, 2048) and ch.dtype == torch.int64
    for b in range(2):
        pix = ch[b, 0]
        assert (d[b].reshape(-1)[pix] > 1e-6).all()            # only valid pixels
        assert pix.unique().numel() == 2048                      # without replacement
        assert not torch.equal(pix, pix.sort().values)           # shuffled
    few = torch.zeros(1, 60, 80, device=device)
    few.view(-1)[:500] = 1.0
    ch = inputs.sample_choose(few, 1024)                          # wrap padding
    assert set(ch.unique().tolist()) == set(range(500))
    counts = torch.bincount(ch.reshape(-1), minlength=500)
    assert int(counts.min()) == 2 and int(counts.max()) == 3     # 1024 = 2 * 500 + 24: every pixel twice, 24 of them 3 times
    with pytest.raises(ValueError):
        inputs.sample_choose(torch.zeros(1, 60, 80, device=device), 128)


def test_sample_points_is_a_uniform_subset_and_reproducible(device, trials=300):
    """Same seed -> same sample; different seeds -> different samples; every valid pixel is (about) equally likely to be
    drawn and to land in the first quarter (the index pyramid's 'random' sub-sampling takes prefixes)."""
    d = torch.from_numpy(np.stack([_depth(5), _depth(6)])).to(device)
    d = torch.nan_to_num(d, nan=0.0, posinf=0.0)
    a = inputs.sample_points(d, 2048, seed=11)
    b = inputs.sample_points(d, 2048, seed=11)
    c = inputs.sample_points(d, 2048, seed=12)
    assert torch.equal(a["choose"], b["choose"]) and not torch.equal(a["choose"], c["choose"])
    assert a["n_valid"].tolist() == [(d[i] > 1e-6).sum().item() for i in range(2)]
    n_valid = int(a["n_valid"][0])
    hits = torch.zeros(120 * 160, device=device)
    first = torch.zeros(120 * 160, device=device)
    for s in range(trials):
        ch = inputs.sample_points(d[:1], 2048, seed=1000 + s)["choose"][0, 0]
        hits[ch] += 1
        first[ch[:512]] += 1
    valid = d[0].reshape(-1) > 1e-6
    assert float(hits[~valid].sum()) == 0
    p = 2048 / n_valid                                               # inclusion probability of a valid pixel
    mean = float(hits[valid].mean()) / trials
    assert abs(mean - p) < 1e-6
    sd = (p * (1 - p) / trials) ** 0.5
    assert float((hits[valid] / trials - p).abs().max()) < 6 * sd    # no pixel is favoured
    assert float((first[valid] / trials - p / 4).abs().max()) < 6 * (p / 4 * (1 - p / 4) / trials) ** 0.5


@pytest.mark.parametrize("h,w,n", [(120, 160, 2048), (480, 640, 12288), (60, 80, 6000)])
def test_sample_points_is_the_documented_permutation(device, h, w, n, frames=3):
    """The sample IS the prefix of the frame's valid pixels sorted stably by the documented 32-bit hash (oracle/inputs_ref.sample_order):
    pins the library's own segmented radix sort (csrc/seg_sort.hip, round 5) at the full image size -- 150 chunks per segment, the
    path with per-segment prefix sums -- at the small size (fused sums) and with wrap-around (more samples than valid pixels)."""
    deps = np.stack([_depth(20 + b, h, w) for b in range(frames)])
    deps = np.nan_to_num(deps, nan=0.0, posinf=0.0)
    if n > h * w:
        deps[:, h // 2:, :] = 0.0
    seed = 0x1234567_89abcdef
    got = inputs.sample_points(torch.from_numpy(deps).to(device), n, seed=seed)
    order, n_valid = inputs_ref.sample_order(deps, seed)
    assert got["n_valid"].cpu().tolist() == n_valid.tolist()
    for b in range(frames):
        want = order[b, np.arange(n) % n_valid[b]]
        assert np.array_equal(got["choose"][b, 0].cpu().numpy(), want), b


@pytest.mark.parametrize("n,h,w", [(12288, 480, 640), (1100, 104, 136), (1000, 50, 70)])
def test_pyramid_point_sets_equal_the_slices(device, n, h, w, frames=3):
    """ffb6d_pyramid_sets_f32 (one launch) against the dataset code's slices (linemod_dataset.py:299-323): cloud rows, prefixes,
    coordinate table, strided xyz grids -- from point-major and from channel-major clouds, sizes the strides do not divide."""
    from ffb6d_amd import pyramid, ops_pm
    rng = np.random.RandomState(n)
    crn = torch.from_numpy(rng.randn(frames, 9, n).astype(np.float32)).to(device)
    dpt = torch.from_numpy(rng.randn(frames, 3, h, w).astype(np.float32)).to(device)
    cld = crn[:, :3, :].transpose(1, 2).contiguous()
    for cm in (True, False):
        sets, table = pyramid.point_sets(crn if cm else cld, dpt, channel_major=cm, with_table=True)
        assert torch.equal(table, ops_pm.xyz_table(cld))
        cur = cld
        for i in range(5):
            assert torch.equal(sets[('c', i)], cur), (cm, i)
            if i < 4:
                cur = cur[:, :cur.shape[1] // pyramid.SUB_RATIO[i], :].contiguous()
        for s in (2, 4, 8):
            assert torch.equal(sets[('g', s)], pyramid.strided_grid(dpt, s)), (cm, s)
    sets, table = pyramid.point_sets(cld, dpt)
    assert table is None and torch.equal(sets[('c', 0)], cld)


def test_assemble_inputs_feeds_the_model(device):
    rng = np.random.RandomState(0)
    deps = torch.from_numpy(np.stack([_depth(7), _depth(8)])).to(device)
    deps = torch.nan_to_num(deps, nan=0.0, posinf=0.0)
    rgb = torch.from_numpy(rng.randint(0, 256, (2, 3, 120, 160)).astype(np.uint8)).to(device)
    nrm = torch.from_numpy(rng.randn(2, 3, 120, 160).astype(np.float32)).to(device)
    d = inputs.assemble_inputs(rgb, deps, nrm, synth.LINEMOD_K, 1024, cam_scale=1000.0)
    assert d['cld_rgb_nrm'].shape == (2, 9, 1024) and d['cld_xyz0'].shape == (2, 1024, 3)
    flat = d['dpt_xyz'].reshape(2, 3, -1)
    idx = d['choose'].expand(2, 3, 1024)
    assert torch.equal(d['cld_rgb_nrm'][:, :3], torch.gather(flat, 2, idx))
    assert torch.equal(d['cld_rgb_nrm'][:, 3:6], torch.gather(rgb.float().reshape(2, 3, -1), 2, idx))      # linemod_dataset.py:285
    assert torch.equal(d['cld_rgb_nrm'][:, 6:9], torch.gather(nrm.reshape(2, 3, -1), 2, idx))
    assert torch.equal(d['cld_xyz0'], d['cld_rgb_nrm'][:, :3].transpose(1, 2))
    for k in ('cld_nei_idx0', 'r2p_ds_nei_idx3', 'p2r_up_nei_idx2'):
        assert k in d


def test_depth_normal_matches_the_restatement_bit_exactly(device):
    """Surface normals (normalSpeed.depth_normal's published algorithm, parity unpinned -- see
    oracle/inputs_ref.py): HIP kernel vs the numpy restatement, bit exact, on depth with holes, steps, far pixels
    (>= distance_threshold) and both input dtypes."""
    rng = np.random.RandomState(4)
    yy, xx = np.mgrid[0:120, 0:160]
    d = (900.0 + 1.5 * xx + 0.7 * yy + 3.0 * rng.randn(120, 160)).astype(np.float32)
    d[30:50, 40:90] += 300.0                       # a step: neighbours beyond the difference threshold are dropped
    d[rng.rand(120, 160) < 0.05] = 0.0              # holes
    d[80:, 100:] = 2500.0                           # beyond the distance threshold
    for K in (synth.LINEMOD_K, np.array([[1066.778, 0., 312.9869], [0., 1067.487, 241.3109], [0., 0., 1.]])):
        want = inputs_ref.depth_normal(d, K[0][0], K[1][1], 5, 2000, 20, False)                  # [H,W,3]
        got = inputs.depth_normal(torch.from_numpy(d).to(device), K[0][0], K[1][1], 5, 2000, 20, False)
        assert got.shape == (3, 120, 160)
        np.testing.assert_array_equal(got.permute(1, 2, 0).cpu().numpy(), want)
        d16 = torch.from_numpy(d.astype(np.uint16).astype(np.int32)).to(torch.int16).to(device)   # same bits as uint16
        got16 = inputs.depth_normal(torch.stack([d16, d16]), K[0][0], K[1][1])
        np.testing.assert_array_equal(got16[1].permute(1, 2, 0).cpu().numpy(), want)
    n = np.linalg.norm(want, axis=2)
    assert ((np.abs(n - 1) < 1e-5) | (n == 0)).all() and (n[80:, 100:] == 0).all() and (n[:5] == 0).all()
    with pytest.raises(Exception):
        inputs.depth_normal(torch.from_numpy(d).to(device), 500.0, 500.0, point_into_surface=True)


def _holey_depth(seed, h=120, w=160, unit=10000.0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    metres = 0.6 + 0.012 * xx + 0.004 * yy + 0.002 * rng.randn(h, w)       # spans the near / medium / far bins (1 m, 2 m)
    metres[40:70, 60:100] += 0.9                                            # an object in front of / behind the plane
    d = (metres * unit).astype(np.uint16).astype(np.float32)
    d[rng.rand(h, w) < 0.25] = 0.0                                          # sensor drop-outs
    d[50:58, 20:45] = 0.0                                                   # a hole larger than the 5x5 kernels
    d[:12] = 0.0                                                            # nothing above the top row
    d[:, 150:] = 0.0                                                        # empty columns
    return d


def test_fill_missing_matches_the_restatement(device):
    """YCB depth hole filling (basic_utils.py:467-487 -> depth_map_utils_ycb.py:290-445; parity
    unpinned, see oracle/holefill_ref.py): HIP chain vs the scipy restatement.  Everything before the bilateral blur is
    max / min / median selection and must agree exactly where the blur is not applied; the blur itself within 1e-5."""
    from oracle import holefill_ref
    deps = np.stack([_holey_depth(1), _holey_depth(2)])
    want = np.stack([holefill_ref.fill_missing(d, 10000.0, 1) for d in deps])
    got = inputs.fill_missing(torch.from_numpy(deps).to(device), 10000.0, 1).cpu().numpy()
    assert got.dtype == np.float32 and got.shape == want.shape
    assert ((got > 0) == (want > 0)).all()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    assert (got[:, :7, :140] == 0).all()                                     # never extrapolates upwards (extrapolate=False)
    assert (got[:, 20:, :140] > 0).all()                                     # every hole below the top row is filled
    one = inputs.fill_missing(torch.from_numpy(deps[1]).to(device), 10000.0, 1).cpu().numpy()
    np.testing.assert_array_equal(one, got[1])
    # other scale arguments (the LineMOD-style millimetre unit, scale_2_80m != 1)
    d_mm = _holey_depth(3, unit=1000.0)
    np.testing.assert_allclose(inputs.fill_missing(torch.from_numpy(d_mm).to(device), 1000.0, 2.0, max_depth=6.0).cpu().numpy(),
                               holefill_ref.fill_missing(d_mm, 1000.0, 2.0, max_depth=6.0), rtol=1e-5, atol=0)
    with pytest.raises(Exception):
        inputs.fill_missing(torch.zeros(1, 4, 4, device=device), 1000.0)


def test_f4_kernels_reproduce_the_committed_vectors(device):
    """tests/golden/f4_vectors.npz (made by tests/golden/make_golden_f4.py from the restatements that tests/test_f4_pin_cpu.py checks
    primitive by primitive and against analytic planes / spheres): normals bit exact, filled depth within 1e-5 (the bilateral blur's
    exponential), also against the table-interpolated form OpenCV's float path uses"""
    import os
    from conftest import GOLDEN
    from oracle import holefill_ref
    g = np.load(os.path.join(GOLDEN, "f4_vectors.npz"))
    n_cases = 0
    for k in [k for k in g.files if k.endswith("/depth_mm")]:
        tag = k.split("/")[0]
        fx, fy = (float(v) for v in g[tag + "/fxfy"])
        got = inputs.depth_normal(torch.from_numpy(g[k]).to(device), fx, fy, 5, 2000, 20, False)
        np.testing.assert_array_equal(got.permute(1, 2, 0).cpu().numpy(), g[tag + "/normals"])
        n_cases += 1
    for k in [k for k in g.files if k.endswith("/depth_raw")]:
        tag = k.split("/")[0]
        raw = g[k].astype(np.float32)
        got = inputs.fill_missing(torch.from_numpy(raw).to(device), float(g[tag + "/cam_scale"]), 1).cpu().numpy()
        want = g[tag + "/filled"]
        assert ((got > 0) == (want > 0)).all()
        # 1e-5 per element, or 1e-6 of the map's range where the final `max_depth - d` cancels (pixels near max_depth)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6 * float(want.max()))
        n_cases += 1
    assert n_cases == 6
