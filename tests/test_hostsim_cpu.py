"""Per-thread kernel bodies of the product library, compiled for the host and run on CPU against torch
(tests/hostsim/): index arithmetic and numerics of kernels without cross-lane operations are checked here,
without a GPU, on the very source the GPU executes."""
import ctypes
import shutil

import numpy as np
import pytest
import torch

from ffb6d_amd import forward_pm, model

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not __import__("os").path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc is needed to compile the host simulation")


@pytest.fixture(scope="module")
def sim():
    from tests.hostsim import build
    lib = ctypes.CDLL(build.build())
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.hostsim_upconv_combine.restype = ctypes.c_int
    lib.hostsim_upconv_combine.argtypes = [ctypes.c_int, vp, vp, ctypes.c_float, vp, i64, i64, i64, i64, i64, i64, ctypes.c_int]
    return lib


def _up_block(cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    ub = model.UpBlock(cin, cout).eval()
    with torch.no_grad():
        for p in ub.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        bn = ub.conv[2]
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
        ub.conv[3].weight.fill_(0.25 + 0.05 * seed)
    return ub


def _folded_on_host(sim, ub, x, dt, blocked=0):
    """the folded up-convolution exactly as forward_pm.up_block runs it, with torch for the GEMM and the host simulation of
    upconv_combine_pm_kernel for the second half; x [B,cin,h,w] float32 -> [B,cout,2h,2w] float32"""
    B, cin, h, w = x.shape
    w9, shift, slope = forward_pm.upconv_folded(ub, dt)
    rows = x.permute(0, 2, 3, 1).contiguous().to(dt)                               # [B,h,w,cin] pixel-major
    z = (rows.float().reshape(-1, cin) @ w9.float().t()).to(dt).reshape(B, h, w, -1).contiguous()
    cout = shift.numel()
    out = torch.empty(B, 2 * h, 2 * w, cout, dtype=dt)
    rc = sim.hostsim_upconv_combine(1 if dt == torch.bfloat16 else 0, z.data_ptr(), shift.data_ptr(), slope, out.data_ptr(),
                                    B, h, w, 2 * h, 2 * w, cout, blocked)
    assert rc == 0
    return out.float().permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,cin,cout,h,w", [(2, 16, 8, 5, 7), (1, 8, 16, 1, 1), (1, 24, 4, 2, 9), (3, 8, 12, 6, 3)])
def test_folded_upconv_equals_upsample_conv_bn_prelu(sim, B, cin, cout, h, w):
    """PSPUpsample (pspnet.py:34-45) = the reference modules run by torch on CPU; bar 1e-5 of the output range."""
    ub = _up_block(cin, cout, seed=B + cin)
    x = torch.randn(B, cin, h, w, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        want = ub.conv(x)
        got = _folded_on_host(sim, ub, x, torch.float32)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-5 * scale
    # the negative (PReLU) side and the border taps are exercised
    assert float(want.min()) < 0 and got.shape == want.shape


@pytest.mark.parametrize("B,cin,cout,h,w", [(2, 16, 8, 5, 6), (1, 8, 16, 1, 2), (1, 24, 4, 2, 10), (3, 8, 12, 7, 4), (1, 8, 8, 9, 14), (1, 8, 8, 60, 80)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("form", [1, 2, 3, 4])
def test_register_blocked_combine_is_bit_identical_to_the_simple_form(sim, B, cin, cout, h, w, dt, form):
    """combine_block_body (2 x 4 output pixels per thread, window of 3 x 4 source pixels per tap; form 1 = operand selects,
    form 2 = compile-time operand pattern per tap where a thread's positions follow it, form 3 = per filter row with the
    three taps' loads issued together) against combine_body: the same operations in
    the same order per output -> equal bits, on maps with every border case (1 .. 9 source rows)."""
    if dt == torch.bfloat16 and cout % 8:
        pytest.skip("bf16 rows come in 8-channel units")
    if form == 4 and dt != torch.bfloat16:
        pytest.skip("form 4 = the block form on half bf16 units (4 channels in 8 bytes: what the launcher runs in bf16 since round 6)")
    ub = _up_block(cin, cout, seed=h + w)
    x = torch.randn(B, cin, h, w, generator=torch.Generator().manual_seed(h * w))
    with torch.no_grad():
        simple = _folded_on_host(sim, ub, x, dt, blocked=0)
        blocked = _folded_on_host(sim, ub, x, dt, blocked=form)
    assert torch.equal(simple, blocked)


def test_register_blocked_combine_keeps_nan_where_the_simple_form_puts_it(sim):
    """a NaN source pixel reaches exactly the outputs whose taps blend it (operands are selected, not weighted by zero)"""
    ub = _up_block(8, 8, seed=5)
    x = torch.randn(1, 8, 6, 8, generator=torch.Generator().manual_seed(1))
    x[0, :, 2, 5] = float("nan")
    with torch.no_grad():
        simple = _folded_on_host(sim, ub, x, torch.float32, blocked=0)
        blocked = _folded_on_host(sim, ub, x, torch.float32, blocked=2)
        assert torch.equal(torch.nan_to_num(blocked), torch.nan_to_num(_folded_on_host(sim, ub, x, torch.float32, blocked=1)))
    assert torch.equal(torch.isnan(simple), torch.isnan(blocked)) and 0 < int(torch.isnan(simple).sum()) < simple.numel() // 4
    assert torch.equal(torch.nan_to_num(simple), torch.nan_to_num(blocked))


def test_folded_upconv_bf16_rows(sim):
    """bfloat16 rows (configuration 5): the same bodies with 8-channel units; bar = bf16 rounding of z and of the output."""
    ub = _up_block(16, 8, seed=3)
    x = torch.randn(2, 16, 4, 6, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        want = ub.conv(x)
        got = _folded_on_host(sim, ub, x, torch.bfloat16)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 3e-2 * scale
    assert float((got - want).abs().mean()) <= 5e-3 * scale


def test_folded_weights_follow_weight_updates():
    """the regrouped weights are cached per module and keyed on the version of every source tensor"""
    ub = _up_block(8, 4, seed=1)
    w9a, sa, _ = forward_pm.upconv_folded(ub)
    assert forward_pm.upconv_folded(ub)[0] is w9a
    with torch.no_grad():
        ub.conv[1].weight.mul_(2.0)
    w9b, sb, _ = forward_pm.upconv_folded(ub)
    assert torch.allclose(w9b, 2 * w9a) and torch.equal(sa, sb)
    # layout: row (ky*3+kx)*cout + co
    cv, bn = ub.conv[1], ub.conv[2]
    scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).detach()
    assert torch.allclose(w9b[(1 * 3 + 2) * 4 + 3], (cv.weight[3, :, 1, 2] * scale[3]).detach())
    assert np.isfinite(w9b.numpy()).all()


# ---------------------------------------------------------------------------------------------------------------
# relative position encoding fused with lfa.mlp1 (csrc/posenc_body.h)
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sim_posenc(sim):
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    sim.hostsim_posenc_mlp.restype = ctypes.c_int
    sim.hostsim_posenc_mlp.argtypes = [i32, vp, vp, i32, vp, i64, vp, i32, vp, i64, i64, i32, i64, i64]
    return sim


@pytest.mark.parametrize("B,N,K,cout,idt,blocks", [(2, 50, 16, 16, torch.int64, 3), (1, 33, 16, 128, torch.int32, 1),
                                                    (3, 17, 5, 24, torch.int64, 6), (1, 1, 1, 8, torch.int64, 2)])
def test_fused_posenc_mlp_equals_encoding_then_shared_mlp(sim_posenc, B, N, K, cout, idt, blocks):
    """RandLANet.py:196-199: mlp1(relative_pos_encoding(xyz, idx)), with the oracle's plain-torch encoding and a float64
    matmul as the reference; bar 1e-5 of the output range; also the padded [cout,16] weight layout the forward passes."""
    from oracle import ops_ref
    g = torch.Generator().manual_seed(B * 100 + N)
    xyz = torch.randn(B, N, 3, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g).to(idt)
    w = torch.zeros(cout, 16)
    w[:, :10] = torch.randn(cout, 10, generator=g) * 0.5
    w[:, 10:] = 7.0                                            # must be ignored
    bias = torch.randn(cout, generator=g)
    enc = ops_ref.relative_pos_encoding(xyz, idx.long())       # [B,N,K,10]
    for act, f in ((0, lambda v: v), (1, torch.relu), (2, lambda v: torch.nn.functional.leaky_relu(v, 0.2))):
        want = f(enc.double() @ w[:, :10].double().t() + bias.double())
        out = torch.full((B, N, K, cout), float("nan"))
        rc = sim_posenc.hostsim_posenc_mlp(0, xyz.data_ptr(), idx.data_ptr(), 64 if idt == torch.int64 else 32, w.data_ptr(), 16,
                                           bias.data_ptr(), act, out.data_ptr(), B, N, K, cout, blocks * (3 if cout == 24 else 1))
        assert rc == 0
        assert float((out.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    outb = torch.zeros((B, N, K, cout), dtype=torch.bfloat16)
    if cout % 8 == 0:
        rc = sim_posenc.hostsim_posenc_mlp(1, xyz.data_ptr(), idx.data_ptr(), 64 if idt == torch.int64 else 32, w.data_ptr(), 16,
                                           bias.data_ptr(), 2, outb.data_ptr(), B, N, K, cout, blocks)
        assert rc == 0
        assert float((outb.double() - want).abs().max()) <= 1e-2 * float(want.abs().max())


def test_static_operand_pattern_covers_the_interior_of_the_real_maps(sim):
    """the compile-time operand pattern of the blocked tap blend must hold for all but the border blocks of the three
    PSPUpsample maps (60x80, 120x160, 240x320 sources): else the fast path would silently never run"""
    sim.hostsim_upconv_static_share.restype = ctypes.c_double
    sim.hostsim_upconv_static_share.argtypes = [ctypes.c_int64, ctypes.c_int64]
    for ih, iw, least in ((60, 80, 0.90), (120, 160, 0.95), (240, 320, 0.97)):
        share = sim.hostsim_upconv_static_share(ih, iw)
        print("static share", ih, iw, share)
        assert share >= least, (ih, iw, share)


@pytest.mark.parametrize("nbx,nby", [(10, 1920), (40, 960), (1, 1), (3, 5), (7, 1), (1, 9), (40, 3840), (2, 4)])
def test_xcd_band_order_is_a_bijection_with_one_contiguous_band_per_xcd(sim, nbx, nby):
    sim.hostsim_xcd_band_check.restype = ctypes.c_int
    sim.hostsim_xcd_band_check.argtypes = [ctypes.c_uint, ctypes.c_uint]
    assert sim.hostsim_xcd_band_check(nbx, nby) == 0
