"""GPU parity of the neighbour operators (through the C ABI) against the reference goldens
and a plain PyTorch fp32 CPU restatement (oracle/ops_ref.py).  Bar: gathers / max pooling
bit-exact; softmax pooling and position encoding within 1e-5 (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import ops
from oracle import ops_ref

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def test_ops_match_reference_goldens(device):
    z = np.load(os.path.join(GOLDEN, "ops_small.npz"))
    t = lambda k: torch.from_numpy(z[k]).to(device)
    for idt in (torch.int64, torch.int32):
        np.testing.assert_array_equal(ops.random_sample(t("feat").unsqueeze(3), t("pool_idx").to(idt)).cpu().numpy(),
                                      z["random_sample"])
        np.testing.assert_array_equal(ops.nearest_interpolation(t("feat").unsqueeze(3), t("interp_idx").to(idt)).cpu().numpy(),
                                      z["nearest_interpolation"])
        np.testing.assert_array_equal(ops.gather_neighbour(t("pc"), t("nei").to(idt)).cpu().numpy(), z["gather_neighbour"])
        rpe = ops.relative_pos_encoding(t("xyz"), t("nei").to(idt)).cpu().numpy()
        np.testing.assert_array_equal(rpe[..., 1:], z["relative_pos_encoding"][..., 1:])
        # the norm column differs from torch-CPU's by <= 1 ulp in ~0.1 % of entries (numpy's
        # own (x+y)+z, sqrt differs from torch in as many); the bar for features is 1e-5
        np.testing.assert_allclose(rpe[..., 0], z["relative_pos_encoding"][..., 0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ops.att_pool(t("fs"), t("act")).cpu().numpy(), z["att_pool"], **TOL)


# (C, M, Np) shapes of FFB6D at N=12288 plus ragged ones
@pytest.mark.parametrize("B,C,M,Np,K", [(2, 64, 12288, 3072, 16), (1, 1024, 4800, 48, 16), (2, 64, 76800, 768, 16),
                                        (1, 5, 37, 300, 16), (3, 7, 100, 1, 4), (1, 512, 192, 48, 16)])
def test_random_sample(device, B, C, M, Np, K):
    g = torch.Generator().manual_seed(C + M + Np)
    feat = torch.randn(B, C, M, generator=g)
    idx = torch.randint(0, M, (B, Np, K), generator=g)
    want = ops_ref.random_sample(feat, idx)
    got = ops.random_sample(feat.to(device).unsqueeze(3), idx.to(device))
    assert got.shape == (B, C, Np, 1)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(ops.random_sample(feat.to(device), idx.to(device).int()).cpu(), want)  # 3-d feature, int32 idx


@pytest.mark.parametrize("B,C,M,U", [(2, 64, 3072, 19200), (1, 1024, 48, 4800), (1, 64, 3072, 76800),
                                     (2, 9, 50, 333), (1, 3, 5, 1), (1, 512, 48, 192)])
def test_nearest_interpolation(device, B, C, M, U):
    g = torch.Generator().manual_seed(C + M + U)
    feat = torch.randn(B, C, M, 1, generator=g)
    idx = torch.randint(0, M, (B, U, 1), generator=g)
    want = ops_ref.nearest_interpolation(feat, idx)
    got = ops.nearest_interpolation(feat.to(device), idx.to(device))
    assert got.shape == (B, C, U, 1)
    assert torch.equal(got.cpu(), want)


def test_choose_gather(device):
    g = torch.Generator().manual_seed(1)
    rgb = torch.randn(2, 64, 30, 40, generator=g)
    choose = torch.randint(0, 1200, (2, 1, 500), generator=g)
    want = torch.gather(rgb.view(2, 64, -1), 2, choose.repeat(1, 64, 1))   # ffb6d.py:309-312
    assert torch.equal(ops.choose_gather(rgb.to(device), choose.to(device)).cpu(), want)


@pytest.mark.parametrize("B,N,C,K", [(2, 3072, 32, 16), (1, 500, 3, 16), (1, 192, 128, 16), (2, 77, 6, 5)])
def test_gather_neighbour(device, B, N, C, K):
    g = torch.Generator().manual_seed(N + C)
    pc = torch.randn(B, N, C, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g)
    got = ops.gather_neighbour(pc.to(device), idx.to(device))
    assert got.shape == (B, N, K, C)
    assert torch.equal(got.cpu(), ops_ref.gather_neighbour(pc, idx))


@pytest.mark.parametrize("B,N,K", [(2, 3072, 16), (1, 100, 16), (3, 33, 7), (1, 12288, 16)])
def test_relative_pos_encoding(device, B, N, K):
    g = torch.Generator().manual_seed(N + K)
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    idx = torch.randint(0, N, (B, N, K), generator=g)
    got = ops.relative_pos_encoding(xyz.to(device), idx.to(device)).cpu()
    want = ops_ref.relative_pos_encoding(xyz, idx)
    assert got.shape == (B, N, K, 10)
    assert torch.equal(got[..., 1:], want[..., 1:])              # differences and copies are exact
    torch.testing.assert_close(got[..., 0], want[..., 0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,C,N,K", [(2, 32, 3072, 16), (1, 256, 192, 16), (1, 5, 33, 16), (2, 4, 50, 8),
                                     (1, 3, 20, 32), (1, 3, 20, 4), (1, 6, 11, 5)])
def test_att_pool(device, B, C, N, K):
    g = torch.Generator().manual_seed(C + N + K)
    fs = torch.randn(B, C, N, K, generator=g)
    act = 4 * torch.randn(B, C, N, K, generator=g)
    got = ops.att_pool(fs.to(device), act.to(device))
    assert got.shape == (B, C, N, 1)
    torch.testing.assert_close(got.cpu(), ops_ref.att_pool(fs, act), **TOL)


def test_backward_matches_torch_autograd(device):
    g = torch.Generator().manual_seed(8)
    B, C, M, Np, K, U = 2, 6, 40, 17, 16, 90
    feat = torch.randn(B, C, M, generator=g)
    pool = torch.randint(0, M, (B, Np, K), generator=g)
    up = torch.randint(0, M, (B, U, 1), generator=g)
    pc = torch.randn(B, M, 8, generator=g)
    nei = torch.randint(0, M, (B, M, K), generator=g)
    fs = torch.randn(B, C, 21, K, generator=g)
    act = torch.randn(B, C, 21, K, generator=g)

    def grads(fn, *leaves):
        ls = [l.clone().requires_grad_(True) for l in leaves]
        out = fn(*ls)
        w = torch.linspace(-1, 1, out.numel(), device=out.device).view_as(out)
        (out * w).sum().backward()
        return [l.grad.cpu() for l in ls]

    d = lambda x: x.to(device)
    (a,), (b,) = grads(lambda f: ops.random_sample(f, d(pool)), d(feat)), grads(lambda f: ops_ref.random_sample(f, pool), feat)
    torch.testing.assert_close(a, b, **TOL)
    (a,), (b,) = grads(lambda f: ops.nearest_interpolation(f.unsqueeze(3), d(up)), d(feat)), \
        grads(lambda f: ops_ref.nearest_interpolation(f.unsqueeze(3), up), feat)
    torch.testing.assert_close(a, b, **TOL)
    (a,), (b,) = grads(lambda p: ops.gather_neighbour(p, d(nei)), d(pc)), grads(lambda p: ops_ref.gather_neighbour(p, nei), pc)
    torch.testing.assert_close(a, b, **TOL)
    a, b = grads(ops.att_pool, d(fs), d(act)), grads(ops_ref.att_pool, fs, act)
    torch.testing.assert_close(a[0], b[0], **TOL)
    torch.testing.assert_close(a[1], b[1], **TOL)


def test_random_sample_propagates_nan_like_torch_max(device):
    """torch.max (ffb6d.py:176) returns NaN when a gathered neighbour is NaN; both channel-major kernels (one lane per
    point, and the 16-lane row kernel used for image sources) must do the same instead of masking it."""
    for M, Np in ((64, 16), (4096, 32)):                 # M <= 4*Np: lane kernel;  M > 4*Np: row16 kernel
        f = torch.zeros(1, 3, M)
        f[0, 1, 5] = float("nan")
        idx = torch.stack([torch.arange(16) + s for s in range(Np)]).view(1, Np, 16) % M
        want = ops_ref.random_sample(f.unsqueeze(3), idx)
        got = ops.random_sample(f.to(device).unsqueeze(3), idx.to(device)).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(want)) and bool(torch.isnan(got).any())
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


def test_model_index_check_catches_out_of_range_entries(device, monkeypatch):
    from ffb6d_amd import model, pyramid, synth
    frames = synth.make_batch(7, 1, n_points=1024, height=120, width=160)
    inputs = pyramid.frames_to_device(frames, device)
    net = model.FFB6D(n_classes=5, n_pts=1024).to(device).eval()
    net.check_indices(inputs)                              # a pyramid built by this package is clean
    inputs['p2r_ds_nei_idx1'] = inputs['p2r_ds_nei_idx1'].clone()
    inputs['p2r_ds_nei_idx1'][0, 3, 0] = 10 ** 6
    with pytest.raises(IndexError, match="p2r_ds_nei_idx1"):
        net.check_indices(inputs)
    monkeypatch.setenv("FFB6D_CHECK_INDICES", "1")
    with torch.no_grad(), pytest.raises(IndexError):
        net(inputs)


def test_index_range_check(device):
    idx = torch.tensor([[0, 5, 9, 10, -1]], device=device)
    assert ops.check_index_range(idx, 10) == 2
    assert ops.check_index_range(idx.int()[:, :3], 10) == 0


@pytest.mark.parametrize("shape,size,ac", [((2, 16, 60, 80), (120, 160), True), ((1, 3, 7, 5), (14, 10), True),
                                           ((2, 8, 6, 6), (60, 80), False), ((1, 4, 1, 1), (60, 80), False),
                                           ((1, 5, 3, 3), (9, 7), False), ((1, 2, 240, 320), (480, 640), True)])
def test_bilinear_resize_matches_torch(device, shape, size, ac):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    want = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=ac)   # CPU ATen
    got = ops.bilinear_resize(x.to(device), size, ac).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def _mlp_ref(x1, w, bias, act, x2=None, gather=None):
    x = x1 if x2 is None else torch.cat([x1, x2], dim=1)
    B, K = x.shape[:2]
    y = torch.einsum("km,bkp->bmp", w.double(), x.reshape(B, K, -1).double())
    if bias is not None:
        y = y + bias.double().view(1, -1, 1)
    if gather is not None:
        Y, idx = gather
        y = y + torch.gather(Y.double(), 2, idx.long().unsqueeze(1).expand(-1, Y.shape[1], -1))
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.nn.functional.leaky_relu(y, 0.2)
    return y.float()


# (B, K1, K2, Cout, P, act, gather) -- shapes of FFB6D's shared MLPs incl. ragged ones
@pytest.mark.parametrize("B,K1,K2,Cout,P,act,py", [
    (2, 9, 0, 8, 12288, 2, 0), (1, 10, 0, 16, 3072 * 16, 2, 0), (2, 32, 0, 32, 768 * 16, 0, 0),
    (2, 64, 64, 64, 3072, 1, 0), (1, 1024, 0, 1024, 4800, 1, 48), (2, 128, 0, 64, 4800, 1, 768),
    (1, 256, 512, 256, 192, 2, 0), (1, 128, 0, 22, 12288, 0, 0), (3, 17, 5, 37, 301, 1, 13),
    (1, 2048, 0, 1024, 640, 1, 0), (2, 128, 0, 128, 130, 2, 0),
    # small per-frame P: flat columns over frames + split-K partial slabs
    (8, 1024, 0, 512, 48, 1, 0), (8, 512, 512, 256, 192, 2, 0), (8, 256, 0, 256, 192, 1, 48),
    (4, 128, 0, 22, 768, 0, 0), (2, 48, 16, 40, 1024, 1, 100),
    # pipelined kernel (per-frame tiles, Cout > 32): K tail (rows past K come back as zeros from the buffer range
    # check), second source shorter than a k-tile, Cout that is no multiple of the row tile, ragged last column tile
    (1, 32, 8, 64, 4096, 1, 0), (1, 48, 0, 72, 2052, 2, 50), (2, 16, 16, 128, 2048, 0, 0),
    (1, 1000, 0, 136, 2048, 1, 0), (2, 64, 24, 132, 2304, 2, 7),
    # a few ragged columns per frame (pyramid-pooling levels 1x1 and 3x3): padded into the flat kernel
    (8, 512, 0, 1024, 1, 0, 0), (8, 512, 0, 1024, 9, 0, 0), (2, 64, 32, 128, 7, 1, 0),
])
def test_shared_mlp_matches_fp64_reference(device, B, K1, K2, Cout, P, act, py):
    g = torch.Generator().manual_seed(K1 + Cout + P)
    x1 = torch.randn(B, K1, P, generator=g)
    x2 = torch.randn(B, K2, P, generator=g) if K2 else None
    w = torch.randn(K1 + K2, Cout, generator=g) / (K1 + K2) ** 0.5
    bias = torch.randn(Cout, generator=g)
    gather = (torch.randn(B, Cout, py, generator=g), torch.randint(0, py, (B, P), generator=g)) if py else None
    want = _mlp_ref(x1, w, bias, act, x2, gather)
    d = lambda t: None if t is None else t.to(device)
    got = ops.shared_mlp(d(x1), d(w), d(bias), act, x2=d(x2),
                         gather=None if gather is None else (d(gather[0]), d(gather[1]))).cpu()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_shared_mlp_ignores_non_finite_neighbours_of_its_operands(device):
    """The pipelined kernel lets out-of-tile loads wrap into neighbouring rows; what it reads there may only
    reach accumulator elements that are never stored.  Operands embedded in NaN-filled storage (batch stride
    larger than K*P, weights inside a larger buffer) must give the same result as clean ones."""
    g = torch.Generator().manual_seed(9)
    B, K, C, P = 2, 40, 72, 2052
    x = torch.randn(B, K, P, generator=g).to(device)
    w = (torch.randn(K, C, generator=g) / K ** 0.5).to(device)
    bias = torch.randn(C, generator=g).to(device)
    want = ops.shared_mlp(x, w, bias, ops.ACT_RELU)
    big = torch.full((B, K + 24, P), float("nan"), device=device)
    big[:, :K] = x
    wbuf = torch.full((K + 16, C), float("nan"), device=device)
    wbuf[:K] = w
    got = ops.shared_mlp(big[:, :K], wbuf[:K], bias, ops.ACT_RELU)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)


def test_shared_mlp_4d_views_and_channel_slices(device):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 48, 100, 16, generator=g).to(device)        # [B,C,N,K] like the LFA tensors
    w = (torch.randn(16, 24, generator=g) / 4).to(device)
    got = ops.shared_mlp(x[:, 16:32], w, None, ops.ACT_LEAKY)       # channel slice, batch stride != K*P
    want = _mlp_ref(x[:, 16:32].cpu(), w.cpu(), None, 2).view(2, 24, 100, 16)
    assert got.shape == (2, 24, 100, 16)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)


def test_channel_major_variants_match_the_reference_layout_ops(device):
    g = torch.Generator().manual_seed(12)
    xyz = (torch.rand(2, 300, 3, generator=g) * 2 - 1).to(device)
    idx = torch.randint(0, 300, (2, 300, 16), generator=g).to(device)
    a = ops.relative_pos_encoding_cm(xyz, idx)
    b = ops.relative_pos_encoding(xyz, idx).permute(0, 3, 1, 2).contiguous()
    assert torch.equal(a, b)
    f1 = torch.randn(2, 8, 77, 16, generator=g).to(device)
    f2 = torch.randn(2, 24, 77, 16, generator=g).to(device)
    act = (3 * torch.randn(2, 32, 77, 16, generator=g)).to(device)
    assert torch.equal(ops.att_pool2(f1, f2, act), ops.att_pool(torch.cat([f1, f2], 1), act))


def test_affine_act_matches_bn_relu_add(device):
    g = torch.Generator().manual_seed(3)
    bn = torch.nn.BatchNorm2d(6).eval()
    bn2 = torch.nn.BatchNorm2d(6).eval()
    for m in (bn, bn2):
        m.weight.data.uniform_(0.5, 1.5, generator=g); m.bias.data.normal_(generator=g)
        m.running_mean.normal_(generator=g); m.running_var.uniform_(0.5, 2, generator=g)
    x = torch.randn(2, 6, 10, 12, generator=g)
    r = torch.randn(2, 6, 10, 12, generator=g)
    want = torch.relu(bn(x) + bn2(r))
    bn, bn2 = bn.to(device), bn2.to(device)
    got = ops.affine_act_(x.to(device).clone(), *ops.bn_fold(bn), act=ops.ACT_RELU, residual=r.to(device),
                          res_affine=ops.bn_fold(bn2))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
    want = torch.nn.functional.leaky_relu(bn.cpu()(x), 0.25)
    got = ops.affine_act_(x.to(device).clone(), *ops.bn_fold(bn.to(device)), act=ops.ACT_LEAKY, slope=0.25)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        bn.weight.mul_(2.0)                       # in-place edit (optimizer step, load_state_dict) invalidates the fold
    s2, _ = ops.bn_fold(bn)
    torch.testing.assert_close(s2, bn.weight * torch.rsqrt(bn.running_var + bn.eps))


def test_channel_log_softmax(device):
    x = torch.randn(2, 64, 30, 40, generator=torch.Generator().manual_seed(1)) * 3
    want = torch.log_softmax(x, dim=1)
    got = ops.channel_log_softmax_(x.to(device).clone()).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_psp_pool_and_prior_sum_match_torch(device):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 5, 60, 80, generator=g)
    sizes = [1, 2, 3, 6]
    got = ops.psp_pool(x.to(device), sizes).cpu()
    want = torch.cat([torch.nn.functional.adaptive_avg_pool2d(x, (s, s)).reshape(2, 5, -1) for s in sizes], dim=2)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    z = torch.randn(2, 7, 50, generator=g)
    got = ops.psp_prior_sum(z.to(device), sizes, (60, 80)).cpu()
    want, off = 0, 0
    for s in sizes:
        want = want + torch.nn.functional.interpolate(z[:, :, off:off + s * s].reshape(2, 7, s, s), size=(60, 80),
                                                      mode="bilinear", align_corners=False)
        off += s * s
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,d1,d2,N", [(2, 16, 16, 3072), (1, 32, 32, 300), (2, 64, 64, 48), (1, 128, 128, 192), (1, 8, 24, 7)])
def test_att_score_pool_equals_unfused_attentive_pooling(device, B, d1, d2, N):
    g = torch.Generator().manual_seed(d1 + N)
    f1 = torch.randn(B, d1, N, 16, generator=g)
    f2 = torch.randn(B, d2, N, 16, generator=g)
    w = torch.randn(d1 + d2, d1 + d2, generator=g) / (d1 + d2) ** 0.5          # fc.weight [out, in]
    fs = torch.cat([f1, f2], dim=1)
    att = torch.einsum("oi,binx->bonx", w.double(), fs.double())
    want = (fs.double() * torch.softmax(att, dim=3)).sum(dim=3, keepdim=True).float()
    got = ops.att_score_pool(f1.to(device), f2.to(device), w.t().contiguous().to(device)).cpu()
    assert got.shape == (B, d1 + d2, N, 1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def _grad_pair(fn_ours, fn_ref, x, extra=()):
    """gradients of (y * r).sum() through our Function and through the torch reference"""
    outs = []
    for fn in (fn_ours, fn_ref):
        xs = x.detach().clone().requires_grad_(True)
        ex = [e.detach().clone().requires_grad_(True) for e in extra]
        y = fn(xs, *ex)
        r = torch.linspace(-1.0, 1.0, y.numel(), device=y.device).reshape(y.shape).to(y.dtype)
        (y.float() * r.float()).sum().backward()
        outs.append((y.detach(), xs.grad, [e.grad for e in ex]))
    return outs


@pytest.mark.parametrize("B,C,IH,IW,fmt", [(2, 8, 5, 7, "cl"), (1, 16, 2, 2, "cl"), (2, 8, 6, 4, "nchw"), (1, 24, 9, 3, "cl")])
def test_upsample_align_gather_backward_matches_aten(device, B, C, IH, IW, fmt):
    """pspnet.py:37-42: bilinear x2 with align_corners; forward = torch's, backward = ffb6d_bilinear_bwd_pm (a gather with ATen's
    source-index arithmetic) against ATen's own scatter backward"""
    F = torch.nn.functional
    g = torch.Generator().manual_seed(IH * IW + C)
    x = torch.randn(B, C, IH, IW, generator=g).to(device)
    if fmt == "cl":
        x = x.contiguous(memory_format=torch.channels_last)
    (y, gx, _), (yr, gr, _) = _grad_pair(lambda t: ops.upsample_align(t, (2 * IH, 2 * IW)),
                                          lambda t: F.interpolate(t, size=(2 * IH, 2 * IW), mode="bilinear", align_corners=True), x)
    assert torch.equal(y, yr)
    torch.testing.assert_close(gx, gr, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape,fmt,dt", [((2, 8, 5, 6), "cl", torch.float32), ((3, 16, 4, 4), "nchw", torch.float32), ((2, 8, 6, 6), "cl", torch.bfloat16)])
@pytest.mark.parametrize("slope", [0.25, 1.5, -0.3])
def test_prelu_with_in_kernel_slope_gradient_matches_torch(device, shape, fmt, dt, slope):
    """single-slope PReLU (pspnet.py:43), any learned slope: y, grad_x and the slope's gradient against torch"""
    F = torch.nn.functional
    g = torch.Generator().manual_seed(int(slope * 100) + shape[1])
    x = torch.randn(*shape, generator=g).to(dt).to(device)
    if fmt == "cl":
        x = x.contiguous(memory_format=torch.channels_last)
    w = torch.tensor([slope], device=x.device)
    (y, gx, (gw,)), (yr, gr, (gwr,)) = _grad_pair(ops.prelu, lambda t, a: F.prelu(t.float(), a).to(t.dtype), x, (w,))
    tol = 1e-6 if dt == torch.float32 else 1e-2
    torch.testing.assert_close(y.float(), yr.float(), rtol=tol, atol=tol)
    torch.testing.assert_close(gx.float(), gr.float(), rtol=tol, atol=tol)
    torch.testing.assert_close(gw, gwr, rtol=1e-3 if dt == torch.float32 else 3e-2, atol=1e-3)


@pytest.mark.parametrize("B,C,M,U,idt", [(2, 8, 50, 400, torch.int64), (3, 70, 17, 90, torch.int32), (1, 4, 3072, 5000, torch.int64)])
def test_nearest_interpolation_backward_privatised_in_lds(device, B, C, M, U, idt):
    """FFB6D.nearest_interpolation (ffb6d.py:179-194) backward: the scatter-add of the pixel gradients, privatised per (frame,
    channel group) in LDS, against torch.gather's autograd"""
    g = torch.Generator().manual_seed(M + U)
    feat = torch.randn(B, C, M, 1, generator=g).to(device)
    idx = torch.randint(0, M, (B, U, 1), generator=g).to(idt).to(device)

    def ref(f):
        return torch.gather(f.squeeze(3), 2, idx.long().reshape(B, 1, U).expand(-1, C, -1)).unsqueeze(3)
    (y, gf, _), (yr, gr, _) = _grad_pair(lambda f: ops.nearest_interpolation(f, idx), ref, feat)
    assert torch.equal(y, yr)
    torch.testing.assert_close(gf, gr, rtol=1e-5, atol=1e-4)
