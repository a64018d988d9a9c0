"""Channels-last training operators (ffb6d_amd/ops_cl.py, csrc/train_rows.hip) against plain torch on the reference's own
formulas (ffb6d.py:159-194, RandLANet.py:216-250): forward values and the gradients autograd derives for the torch formulas.
float32: 1e-5 of range (float atomics reorder sums); bfloat16 rows: one bf16 rounding of the result (2^-8 relative).
(Named to run after the other GPU test files: it also holds the cases of the forms written after round 3's last GPU call -- row
LogSoftmax, multi-lane gather backward -- which the device has not executed yet.)"""
import pytest
import torch

from ffb6d_amd import ops_cl

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16]


def cl(x):
    """a [B,C,N,K] tensor in channels_last memory (what a channels_last convolution writes)"""
    return x.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else x


def close(got, want, dt, what, f32_bar=1e-5):
    scale = float(want.abs().max()) + 1e-12
    err = float((got.float() - want.float()).abs().max()) / scale
    assert err <= (f32_bar if dt == torch.float32 else 1.2e-2), (what, err)


def ref_gather(feature, idx):                     # ffb6d.py:179-194 on [B,C,M,1], idx [B,U]
    B, C = feature.shape[:2]
    return torch.gather(feature.reshape(B, C, -1), 2, idx.unsqueeze(1).expand(-1, C, -1).long())


def ref_random_sample(feature, pool_idx):         # ffb6d.py:159-177
    B, C = feature.shape[:2]
    Np, K = pool_idx.shape[1:]
    g = torch.gather(feature.reshape(B, C, -1), 2, pool_idx.reshape(B, 1, Np * K).expand(-1, C, -1).long())
    return g.reshape(B, C, Np, K).max(dim=3, keepdim=True)[0]


def ref_att_pool(feat, scores):                   # RandLANet.py:245-248
    return (feat * torch.softmax(scores, dim=3)).sum(dim=3, keepdim=True)


def grads(fn, *xs, rdt=None):
    xs = [x.detach().clone().requires_grad_(True) for x in xs]
    y = fn(*xs)
    r = torch.linspace(-1.0, 1.0, y.numel(), device=y.device).reshape(y.shape)
    if rdt is not None:            # upstream gradient representable in the operator's output dtype, for both sides
        r = r.to(rdt).float()
    (y.float() * r).sum().backward()
    return [y.detach()] + [x.grad for x in xs]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,C,M,U,idt", [(2, 16, 50, 400, torch.int64), (3, 72, 17, 90, torch.int32), (1, 8, 300, 300, torch.int64)])
def test_nearest_interpolation_rows(device, dt, B, C, M, U, idt):
    g = torch.Generator().manual_seed(M * U + C)
    feat = cl(torch.randn(B, C, M, 1, generator=g)).to(device).to(dt)
    idx = torch.randint(0, M, (B, U, 1), generator=g, dtype=idt).to(device)
    got = grads(lambda f: ops_cl.nearest_interpolation(f, idx), feat)
    want = grads(lambda f: ref_gather(f, idx.reshape(B, U)).unsqueeze(3), feat.float())
    assert got[0].shape == (B, C, U, 1) and got[0].dtype == dt and got[1].dtype == dt
    assert torch.equal(got[0].float(), want[0])                      # a gather moves bits
    close(got[1], want[1], dt, "grad")


@pytest.mark.parametrize("dt", DTS)
def test_gather_neighbour_rows_and_the_gradient_of_a_cat_half(device, dt):
    """[B,C,N,K] output in channels_last memory; the gradient arrives as a channel slice of the concatenation's gradient (row
    stride = both halves) and is consumed without a copy"""
    B, C, N, K = 2, 16, 77, 16
    g = torch.Generator().manual_seed(5)
    feat = cl(torch.randn(B, C, N, 1, generator=g)).to(device).to(dt)
    other = cl(torch.randn(B, 24, N, K, generator=g)).to(device).to(dt)
    idx = torch.randint(0, N, (B, N, K), generator=g).to(device)
    out = ops_cl.gather_neighbour(feat, idx)
    assert out.shape == (B, C, N, K) and out.is_contiguous(memory_format=torch.channels_last)
    got = grads(lambda f: torch.cat([ops_cl.gather_neighbour(f, idx), other], dim=1), feat)
    want = grads(lambda f: torch.cat([ref_gather(f, idx.reshape(B, N * K)).reshape(B, C, N, K), other.float()], dim=1), feat.float())
    assert torch.equal(got[0].float(), want[0])
    close(got[1], want[1], dt, "grad")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,C,M,Np,K", [(2, 16, 200, 50, 16), (1, 64, 48, 12, 16), (2, 8, 30, 30, 5)])
def test_random_sample_rows(device, dt, B, C, M, Np, K):
    g = torch.Generator().manual_seed(M + Np)
    feat = cl(torch.randn(B, C, M, 1, generator=g)).to(device).to(dt)
    idx = torch.randint(0, M, (B, Np, K), generator=g).to(device)
    got = grads(lambda f: ops_cl.random_sample(f, idx), feat)
    want = grads(lambda f: ref_random_sample(f, idx), feat.float())
    assert got[0].shape == (B, C, Np, 1)
    assert torch.equal(got[0].float(), want[0])
    close(got[1], want[1], dt, "grad")          # ties (duplicate indices in a neighbourhood) share one source row: same sum
    # a pixel map as the source (r2p fusion): [B,C,H,W] channels_last, no reshape copy
    fmap = cl(torch.randn(B, C, 6, M // 6 + 1, generator=g)).to(device).to(dt)
    idx2 = torch.randint(0, fmap.shape[2] * fmap.shape[3], (B, Np, K), generator=g).to(device)
    assert torch.equal(ops_cl.random_sample(fmap, idx2).float(), ref_random_sample(fmap.float(), idx2))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,C,N,K", [(2, 32, 100, 16), (1, 256, 12, 16), (2, 8, 33, 7), (1, 16, 5, 1)])
def test_att_pool_rows(device, dt, B, C, N, K):
    g = torch.Generator().manual_seed(C + N)
    feat = cl(torch.randn(B, C, N, K, generator=g)).to(device).to(dt)
    sc = cl(3 * torch.randn(B, C, N, K, generator=g)).to(device).to(dt)
    got = grads(ops_cl.att_pool, feat, sc)
    want = grads(ref_att_pool, feat.float(), sc.float())
    assert got[0].shape == (B, C, N, 1) and got[0].dtype == dt
    for a, b, what in zip(got, want, ("out", "grad_feat", "grad_scores")):
        close(a, b, dt, what)


def test_encoding_rows_are_the_reference_channels_plus_zero_padding(device):
    from ffb6d_amd import ops
    g = torch.Generator().manual_seed(12)
    xyz = (torch.rand(2, 300, 3, generator=g) * 2 - 1).to(device)
    idx = torch.randint(0, 300, (2, 300, 16), generator=g).to(device)
    enc = ops_cl.relative_pos_encoding(xyz, idx)
    assert enc.shape == (2, 16, 300, 16) and enc.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(enc[:, :10], ops.relative_pos_encoding_cm(xyz, idx)) and not enc[:, 10:].any()


def test_odd_channel_counts_fall_back_to_the_channel_major_operators(device):
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(2, 6, 40, 1, generator=g).to(device)
    idx = torch.randint(0, 40, (2, 25, 1), generator=g).to(device)
    assert torch.equal(ops_cl.nearest_interpolation(feat, idx), ref_gather(feat, idx.reshape(2, 25)).unsqueeze(3))
    pool = torch.randint(0, 40, (2, 10, 4), generator=g).to(device)
    assert torch.equal(ops_cl.random_sample(feat, pool), ref_random_sample(feat, pool))


def test_pyramid_pooling_training_fold_on_the_device(device, monkeypatch):
    """model.PyramidPooling: the folded training forward (bin-membership product, per-bin GEMMs, bilinear-weight product added
    onto the x term) against the module as upstream writes it (pspnet.py:7-31), channels_last input.  fp32: same output and
    gradients (1e-4 of range).  Under torch.autocast(bfloat16) both formulations round in different places (and flip ReLU masks
    where the output is ~0), so each is compared with the fp32 result: the fold may not be further away than 2x the upstream
    formulation's own bf16 error + 2e-2 of range, per tensor (measured on CPU autocast: equal or smaller everywhere -- output
    4e-3, input gradient 8e-2 for both, parameters 2e-3 .. 1.6e-2; a wrong weight block or bin is an O(1) error)."""
    from ffb6d_amd import model as M
    torch.manual_seed(3)
    pp = M.PyramidPooling(64, 96).to(device).to(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 15, 20, device=device).relu_().contiguous(memory_format=torch.channels_last)
    r = torch.rand(2, 15, 20, 96, generator=torch.Generator().manual_seed(1)).permute(0, 3, 1, 2).to(device)
    dev_type = "cuda" if torch.cuda.is_available() else "cpu"           # --emulate: CPU tensors, CPU autocast

    def run(fold, autocast):
        monkeypatch.setattr(M.PyramidPooling, "fold_in_training", fold == "1")
        pp.zero_grad()
        xs = x.clone().requires_grad_(True)
        with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=autocast):
            y = pp(xs)
        (y.float() * r).sum().backward()
        return [y.detach().float(), xs.grad] + [p.grad.clone() for p in pp.parameters()]

    def err(a, b):
        return float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)

    truth = run("0", False)
    for a, b in zip(truth, run("1", False)):
        assert err(a, b) <= 1e-4
    for a, b, c in zip(truth, run("0", True), run("1", True)):
        assert err(a, c) <= 2 * err(a, b) + 2e-2, (tuple(a.shape), err(a, c), err(a, b))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 64, 9, 7), (1, 32, 5, 5), (3, 8, 4, 3), (1, 256, 3, 2), (2, 24, 4, 4)])
def test_channel_log_softmax_rows(device, dt, shape):
    """`final` of the colour decoder (pspnet.py:108-112): log_softmax over dim 1 in the map's own dtype, forward and gradient
    against torch in float32 (24 channels: not a power-of-two number of units -> torch fallback, same bars)"""
    g = torch.Generator().manual_seed(sum(shape))
    x = cl(3 * torch.randn(*shape, generator=g)).to(device).to(dt)
    # (the gradient sums 256 upstream values that nearly cancel: both sides get the same bf16-representable upstream gradient)
    got = grads(ops_cl.channel_log_softmax, x, rdt=dt)
    want = grads(lambda t: torch.log_softmax(t, dim=1), x.float(), rdt=dt)
    assert got[0].dtype == dt and got[0].shape == shape
    if shape[1] != 24:
        assert got[0].is_contiguous(memory_format=torch.channels_last)
    close(got[0], want[0], dt, "out")
    close(got[1], want[1], dt, "grad", f32_bar=1e-4)     # g - softmax * sum(g): the 256-term sum is ordered differently (butterfly)


@pytest.mark.parametrize("dt", DTS)
def test_gather_neighbour_with_the_reference_signature_on_rows(device, dt):
    """Building_block.gather_neighbour as the reference calls it (RandLANet.py:225-234): pc [B,N,C], idx [B,N,K] -> [B,N,K,C];
    values, dtype and gradient against torch.gather"""
    B, N, C, K = 2, 60, 16, 16
    g = torch.Generator().manual_seed(9)
    pc = torch.randn(B, N, C, generator=g).to(device).to(dt)
    idx = torch.randint(0, N, (B, N, K), generator=g).to(device)

    def ref(p):                        # index.repeat + gather + reshape, as upstream
        flat = idx.reshape(B, -1, 1).expand(-1, -1, C)
        return torch.gather(p, 1, flat).reshape(B, N, K, C)
    got = grads(lambda p: ops_cl.gather_neighbour_rows(p, idx), pc)
    want = grads(ref, pc.float())
    assert got[0].shape == (B, N, K, C) and got[0].dtype == dt and got[0].is_contiguous()
    assert torch.equal(got[0].float(), want[0])
    close(got[1], want[1], dt, "grad")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("C", [8, 24, 64, 256, 512])
def test_gather_backward_with_skewed_reader_counts(device, dt, C):
    """gather_sum_rows: 1 .. 8 lanes share a (destination row, unit) depending on C (butterfly reduction), rows with no reader at
    all, one row read by a third of the outputs, odd list lengths -- against torch's scatter-add of the same gradient"""
    B, M, U = 2, 37, 301
    g = torch.Generator().manual_seed(C)
    idx = torch.randint(3, M, (B, U), generator=g)          # rows 0..2 have no reader
    idx[:, ::3] = 5                                          # ~100 readers of row 5
    idx[1, 1::7] = M - 1
    feat = cl(torch.randn(B, C, M, 1, generator=g)).to(device).to(dt)
    idx = idx.to(device)
    got = grads(lambda f: ops_cl.nearest_interpolation(f, idx.unsqueeze(2)), feat, rdt=dt)
    want = grads(lambda f: ref_gather(f, idx).unsqueeze(3), feat.float(), rdt=dt)
    assert torch.equal(got[0].float(), want[0])
    assert not got[1][:, :, :3].any()
    close(got[1], want[1], dt, "grad")


def test_empty_index_sets(device):
    """zero outputs: forward shapes, and a zero gradient for the sources (the reference's torch.gather / max formulas do the same)"""
    feat = cl(torch.randn(2, 16, 9, 1)).to(device).requires_grad_(True)
    out = ops_cl.nearest_interpolation(feat, torch.zeros(2, 0, 1, dtype=torch.int64, device=device))
    assert out.shape == (2, 16, 0, 1)
    out.sum().backward()
    assert feat.grad.shape == feat.shape and not feat.grad.any()
    feat.grad = None
    out = ops_cl.random_sample(feat, torch.zeros(2, 0, 16, dtype=torch.int64, device=device))
    assert out.shape == (2, 16, 0, 1)
    out.sum().backward()
    assert not feat.grad.any()
    empty = cl(torch.randn(2, 16, 0, 16)).to(device)
    assert ops_cl.att_pool(empty, empty).shape == (2, 16, 0, 1)
