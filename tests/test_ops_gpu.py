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


def test_channel_major_encoding_matches_the_reference_layout_op(device):
    g = torch.Generator().manual_seed(12)
    xyz = (torch.rand(2, 300, 3, generator=g) * 2 - 1).to(device)
    idx = torch.randint(0, 300, (2, 300, 16), generator=g).to(device)
    a = ops.relative_pos_encoding_cm(xyz, idx)
    b = ops.relative_pos_encoding(xyz, idx).permute(0, 3, 1, 2).contiguous()
    assert torch.equal(a, b)


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
    """pspnet.py:37-42: bilinear x2 with align_corners; forward = the row kernel of the inference path on channels_last maps (ATen's
    source index and blend order: 1e-6 against torch), torch's own elsewhere; backward = ffb6d_bilinear_bwd_pm (a gather with ATen's
    source-index arithmetic) against ATen's own scatter backward"""
    F = torch.nn.functional
    g = torch.Generator().manual_seed(IH * IW + C)
    x = torch.randn(B, C, IH, IW, generator=g).to(device)
    if fmt == "cl":
        x = x.contiguous(memory_format=torch.channels_last)
    (y, gx, _), (yr, gr, _) = _grad_pair(lambda t: ops.upsample_align(t, (2 * IH, 2 * IW)),
                                          lambda t: F.interpolate(t, size=(2 * IH, 2 * IW), mode="bilinear", align_corners=True), x)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last) == yr.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y, yr, rtol=1e-6, atol=1e-6)
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
