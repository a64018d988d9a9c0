"""CPU suite: the module tree is checkpoint-compatible with the reference (names + shapes),
BatchNorm folding is exact algebra, and the patcher swaps the reference's operators."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import model as M
from ffb6d_amd import synth


@pytest.fixture(scope="module")
def ref_shapes():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        return json.load(fh)


def test_state_dict_keys_and_shapes_equal_the_reference(ref_shapes):
    net = M.FFB6D(n_classes=22, n_pts=12288)
    sd = net.state_dict()
    assert set(sd.keys()) == set(ref_shapes.keys())
    for k, v in sd.items():
        assert list(v.shape) == ref_shapes[k], k
    # README: 33.8 M parameters; SURVEY section 6 counted 33,850,392 for the 2-class
    # (LineMOD) model; every extra class adds 128 weights + 1 bias to the segmentation head
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == 33850392 + 20 * 129


def test_reference_checkpoint_loads_strict(ref_shapes):
    net = M.FFB6D(n_classes=22, n_pts=12288)
    sd = synth.synth_state_dict_from_shapes(ref_shapes, seed=0)
    net.load_state_dict(sd, strict=True)
    # and the module-walking generator agrees with the shape-walking one (alias handling)
    sd2 = synth.synth_state_dict(net, seed=0)
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k


@pytest.mark.parametrize("flavour,dims", [("randla", 2), ("pvn", 2), ("pvn", 1)])
def test_batchnorm_folding_is_exact_algebra(flavour, dims):
    torch.manual_seed(0)
    mlp = M.SharedMLP(12, 7, dims=dims, flavour=flavour).eval()
    bn = mlp._bn_module()
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 2.0)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    x = torch.randn(3, 12, 50) if dims == 1 else torch.randn(3, 12, 50, 4)
    with torch.enable_grad():
        want = mlp(x.clone())                 # unfused conv -> BN -> act
    from ffb6d_amd import forward_pm
    w, b = forward_pm.folded(mlp)             # what the fused GPU kernels consume: conv-layout [Cout,Cin], [Cout]
    assert w.shape == (7, 12) and b.shape == (7,)
    y = torch.einsum("mk,bkp->bmp", w, x.reshape(3, 12, -1)) + b.view(1, -1, 1)
    y = torch.nn.functional.leaky_relu(y, 0.2) if flavour == "randla" else torch.relu(y)
    torch.testing.assert_close(y.view_as(want), want, rtol=1e-5, atol=1e-5)
    w16, b16 = forward_pm.folded(mlp, pad_k=16)       # K-padded with zero columns, same bias
    assert w16.shape == (7, 16) and torch.equal(w16[:, :12], w) and (w16[:, 12:] == 0).all() and torch.equal(b16, b)


def test_folded_weight_caches_follow_in_place_edits_and_load_state_dict():
    """Every inference-time weight cache is keyed on the version / storage / device of its sources
    (forward_pm.cached): optimizer-style in-place edits, load_state_dict into a model that already ran in eval(),
    and BatchNorm statistic updates must all be picked up without a train() round trip."""
    from ffb6d_amd import forward_pm
    torch.manual_seed(1)
    mlp = M.SharedMLP(8, 4, flavour="pvn").eval()
    wt0, b0 = forward_pm.folded(mlp)
    assert forward_pm.folded(mlp)[0] is wt0                         # cache hit
    with torch.no_grad():
        mlp.conv.weight.mul_(2.0)
    wt1, _ = forward_pm.folded(mlp)
    torch.testing.assert_close(wt1, 2 * wt0)
    with torch.no_grad():
        mlp._bn_module().running_mean.add_(1.0)
    assert not torch.equal(forward_pm.folded(mlp)[1], b0)
    other = M.SharedMLP(8, 4, flavour="pvn").eval()
    mlp.load_state_dict(other.state_dict())
    torch.testing.assert_close(forward_pm.folded(mlp)[0], forward_pm.folded(other)[0])
    att = M.AttPooling(8, 4).eval()
    w_a = forward_pm.fc_weight(att).clone()
    att.load_state_dict(M.AttPooling(8, 4).state_dict())
    assert not torch.equal(forward_pm.fc_weight(att), w_a)
    blk = M.DilatedResBlock(8, 16).eval()
    net_like = torch.nn.ModuleList([blk])
    net_like.train()                                                # FFB6D.train() analogue is covered on the GPU
    assert blk.training


@pytest.mark.reference
def test_patcher_swaps_the_real_reference_classes():
    """patch_reference on the reference's own modules (models/ffb6d.py, models/RandLA/RandLANet.py): the five members
    are ours afterwards, keep the reference's call signatures, and undo() restores the originals."""
    import inspect
    from ffb6d_amd import ops, patch
    from oracle import ref_harness as rh
    m_ffb6d, m_randla, _ = rh.reference_modules()
    orig = {"rs": m_ffb6d.FFB6D.__dict__["random_sample"], "ni": m_ffb6d.FFB6D.__dict__["nearest_interpolation"],
            "gn": m_randla.Building_block.__dict__["gather_neighbour"],
            "rpe": m_randla.Building_block.__dict__["relative_pos_encoding"], "att": m_randla.Att_pooling.__dict__["forward"]}
    sig = lambda f: list(inspect.signature(f).parameters)           # noqa: E731
    want_sig = {"rs": sig(orig["rs"].__func__), "ni": sig(orig["ni"].__func__), "gn": sig(orig["gn"].__func__),
                "rpe": sig(orig["rpe"]), "att": sig(orig["att"])}
    undo = patch.patch_reference(m_ffb6d, m_randla)
    try:
        assert m_ffb6d.FFB6D.random_sample is ops.random_sample
        assert m_ffb6d.FFB6D.nearest_interpolation is ops.nearest_interpolation
        assert m_randla.Building_block.gather_neighbour is ops.gather_neighbour
        assert m_randla.Building_block.relative_pos_encoding is patch._relative_pos_encoding
        assert m_randla.Att_pooling.forward is patch._att_pooling_forward
        if hasattr(m_randla, "Network"):
            assert m_randla.Network.random_sample is ops.random_sample
        got_sig = {"rs": sig(ops.random_sample), "ni": sig(ops.nearest_interpolation), "gn": sig(ops.gather_neighbour),
                   "rpe": sig(patch._relative_pos_encoding), "att": sig(patch._att_pooling_forward)}
        assert got_sig == want_sig
        # an instance built after patching dispatches to our operators: on CPU tensors they refuse loudly
        from ffb6d_amd import _lib
        with pytest.raises(_lib.FFB6DNativeError):
            m_ffb6d.FFB6D.random_sample(torch.zeros(1, 4, 10, 1), torch.zeros(1, 3, 16, dtype=torch.int64))
    finally:
        undo()
    assert m_ffb6d.FFB6D.__dict__["random_sample"] is orig["rs"]
    assert m_randla.Att_pooling.__dict__["forward"] is orig["att"]


def test_patcher_swaps_reference_operators():
    from ffb6d_amd import ops, patch

    class FakeFFB6D:
        random_sample = staticmethod(lambda f, i: "ref")
        nearest_interpolation = staticmethod(lambda f, i: "ref")

    class FakeBB:
        gather_neighbour = staticmethod(lambda p, i: "ref")

        def relative_pos_encoding(self, xyz, idx):
            return "ref"

    class FakeAtt:
        def forward(self, x):
            return "ref"

    undo = patch.patch_classes(FakeFFB6D, FakeBB, FakeAtt)
    assert FakeFFB6D.random_sample is ops.random_sample
    assert FakeFFB6D.nearest_interpolation is ops.nearest_interpolation
    assert FakeBB.gather_neighbour is ops.gather_neighbour
    assert FakeBB().relative_pos_encoding.__func__ is patch._relative_pos_encoding
    assert FakeAtt().forward.__func__ is patch._att_pooling_forward
    undo()
    assert FakeFFB6D.random_sample(None, None) == "ref" and FakeAtt().forward(None) == "ref"


def test_forward_on_cpu_tensors_fails_loudly():
    """There is no CPU fallback of the hot path: a forward on CPU tensors stops at the first hand-written
    operator with FFB6DNativeError instead of silently computing something with stock torch."""
    import numpy as np
    from ffb6d_amd import _lib, synth
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    frames = synth.make_batch(7, 1, n_points=1024, height=120, width=160)
    idx = opyr.build_batch(frames, oknn.knn_search)
    inputs = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in idx.items()}
    inputs = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in inputs.items()}
    inputs.update(rgb=torch.from_numpy(frames["rgb"]).float(), cld_rgb_nrm=torch.from_numpy(frames["cld_rgb_nrm"]),
                  choose=torch.from_numpy(frames["choose"]).long())
    net = M.FFB6D(n_classes=5, n_pts=1024).eval()
    with torch.no_grad(), pytest.raises(_lib.FFB6DNativeError):
        net(inputs)


def test_upconv_fold_policy(monkeypatch):
    """forward_pm.UPCONV_FOLD: "auto" (the default: every block in fp32; in bf16 by UPCONV_FOLD_BF16, on since round 6) / None (every
    block) / a set of input widths"""
    import torch
    from ffb6d_amd import forward_pm
    assert forward_pm.UPCONV_FOLD == "auto" and forward_pm.UPCONV_FOLD_BF16 is True
    monkeypatch.setattr(forward_pm, "UPCONV_FOLD_BF16", False)
    assert forward_pm._fold_block(1024, torch.bfloat16) is False and forward_pm._fold_block(1024, torch.float32) is True
    monkeypatch.setattr(forward_pm, "UPCONV_FOLD_BF16", True)
    for setting, want32, want16, want_other in (("auto", True, True, True), (frozenset(), False, False, False),
                                                (None, True, True, True), (frozenset((1024, 256)), True, True, False)):
        monkeypatch.setattr(forward_pm, "UPCONV_FOLD", setting)
        assert forward_pm._fold_block(1024, torch.float32) is want32
        assert forward_pm._fold_block(1024, torch.bfloat16) is want16
        assert forward_pm._fold_block(64, torch.float32) is want_other


def test_upconv_folded_without_conv_bias_and_with_per_channel_prelu():
    import pytest
    import torch
    from ffb6d_amd import forward_pm, model
    ub = model.UpBlock(8, 4).eval()
    ub.conv[1] = torch.nn.Conv2d(8, 4, 3, padding=1, bias=False)
    w9, shift, slope = forward_pm.upconv_folded(ub)
    bn = ub.conv[2]
    assert w9.shape == (36, 8) and torch.allclose(shift, (bn.bias - bn.running_mean * bn.weight * torch.rsqrt(bn.running_var + bn.eps)).detach())
    ub.conv[3] = torch.nn.PReLU(4)
    with pytest.raises(NotImplementedError):
        forward_pm.upconv_folded(ub)


@pytest.mark.parametrize("h,w", [(15, 20), (7, 9), (60, 80)])
def test_pyramid_pooling_training_fold_is_the_same_module(h, w):
    """model.PyramidPooling.forward_folded (the training path on the GPU) against the module as upstream writes it (pspnet.py:7-31):
    the constant operators are ATen's own pooling bins / bilinear weights, the output and every gradient agree in fp32."""
    F = torch.nn.functional
    torch.manual_seed(h * w)
    sizes = (1, 2, 3, 6)
    ind, inv, up = M._psp_operators(h, w, sizes, torch.device("cpu"))
    probe = torch.randn(2, 3, h, w)
    rows = probe.permute(0, 2, 3, 1).reshape(2, h * w, 3)
    off = 0
    for s in sizes:
        want = F.adaptive_avg_pool2d(probe, s).permute(0, 2, 3, 1).reshape(2, s * s, 3)
        got = (ind[off:off + s * s] @ rows) * inv[off:off + s * s].view(1, -1, 1)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
        grid = torch.randn(2, 3, s, s)
        want = F.interpolate(grid, size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(2, h * w, 3)
        got = up[:, off:off + s * s] @ grid.permute(0, 2, 3, 1).reshape(2, s * s, 3)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
        off += s * s
    pp = M.PyramidPooling(16, 24, sizes)
    outs = []
    for folded in (False, True):
        pp.zero_grad()
        x = torch.randn(2, 16, h, w, generator=torch.Generator().manual_seed(3)).contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        y = pp.forward_folded(x) if folded else pp(x)
        r = torch.linspace(-1, 1, y.numel()).view(2, h, w, -1).permute(0, 3, 1, 2)
        (y * r).sum().backward()
        outs.append([y.detach(), x.grad] + [p.grad.clone() for p in pp.parameters()])
    for a, b in zip(*outs):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5 * float(a.abs().max()) + 1e-6)
