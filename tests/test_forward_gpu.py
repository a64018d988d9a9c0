"""GPU parity of the whole forward pass: ffb6d_amd.model.FFB6D (HIP kernels + on-device KNN
pyramid) against the end_points the unmodified reference FFB6D produced on CPU for the same
synthetic frames and synthetic weights (tests/golden/forward_*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ffb6d_amd import model as M
from ffb6d_amd import pyramid, synth

pytestmark = pytest.mark.gpu

# Two bars:
#  * HOT PATH: our forward against the plain-torch restatement (oracle/forward_ref.py) run on
#    the SAME device, so both use the same MIOpen convolutions and differ only in the hand-written
#    kernels, the on-device KNN and the BatchNorm-folded GEMMs -> 1e-5 of the output range;
#  * END TO END against the reference's CPU result: dominated by MIOpen-vs-MKLDNN fp32
#    convolution algorithms over ~110 layers (the plain-torch GPU run shows the same
#    deviation, asserted below), so the bar is looser and the measured value is printed.
HOT_TOL = 1e-5
E2E_TOL = 3e-4      # measured 1e-5 .. 1e-4 (printed by the tests)


def build(n_classes, n_pts, device):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        shapes = json.load(fh)
    net = M.FFB6D(n_classes=n_classes, n_pts=n_pts)
    net.load_state_dict(synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=n_classes))
    return net.to(device).eval()


def rel_err(got, want):
    """max |got - want| relative to the range of THIS tensor (no floor: the small-magnitude offset heads are judged
    on their own scale, not on the logits')."""
    scale = float(np.abs(want).max())
    assert scale > 0
    return float(np.abs(got - want).max()) / scale


def assert_close_scaled(got, want, tol, what):
    """per-tensor range bar `tol` plus an elementwise bar: |diff| <= 10*tol*|want| + tol*range."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = float(np.abs(want).max())
    assert scale > 0, what
    diff = np.abs(got - want)
    err = float(diff.max()) / scale
    print(what, "max err / range", err)
    assert err <= tol, (what, err)
    assert (diff <= 10 * tol * np.abs(want) + tol * scale).all(), what


def oracle_on_device(net, inputs):
    from oracle import forward_ref
    sd = {k: v for k, v in net.state_dict().items()}
    with torch.no_grad():
        return forward_ref.ffb6d_forward(sd, {k: (v.long() if v.dtype == torch.int32 else v) for k, v in inputs.items()})


# (2, 8, ...) and (4, 8, ...) are the benchmarked configurations (BASELINE.json configs 2 and 4: bs=8, N=12288 / N=24576); N=12800 is
# the reference's own default geometry (common.py:61 n_sample_points -> 12800 / 3200 / 800 / 200 / 50 points, ragged against every
# tile size of the hand-written kernels), at one frame and at the benchmarked batch; 1100 points on 136 x 168 is the small ragged case
@pytest.mark.parametrize("cfg", [(7, 2, 1024, 120, 160, 5), (1, 1, 12288, 480, 640, 22), (4, 1, 24576, 480, 640, 22),
                                 (2, 8, 12288, 480, 640, 22), (4, 8, 24576, 480, 640, 22),
                                 (2, 1, 12800, 480, 640, 22), (2, 8, 12800, 480, 640, 22), (7, 3, 1100, 136, 168, 5)])
def test_hot_path_matches_plain_torch_on_the_same_device(device, cfg):
    config, bs, n_pts, h, w, n_cls = cfg
    frames = synth.make_batch(config, bs, n_points=n_pts, height=h, width=w)
    net = build(n_cls, n_pts, device)
    inputs = pyramid.frames_to_device(frames, device)
    with torch.no_grad():
        ep = net(inputs)
    ref = oracle_on_device(net, inputs)
    for k in ("pred_rgbd_segs", "pred_kp_ofs", "pred_ctr_ofs"):
        assert_close_scaled(ep[k].cpu().numpy(), ref[k].cpu().numpy(), HOT_TOL, (cfg, k))


@pytest.mark.parametrize("cfg", [(7, 2, 1024, 120, 160, 5), (1, 2, 12288, 480, 640, 22), (2, 2, 12800, 480, 640, 22)])
def test_every_fusion_stage_matches_plain_torch(device, cfg):
    """Stage-level parity of the fused point-major path: both embeddings after each of the 4 encoder and 3 decoder
    fusion stages (ffb6d.py:245-263,281-298) against the plain-torch restatement on the same device, each on the
    range of its own tensor (a whole-network bar alone would hide an O(1)-wrong sub-stage behind later layers)."""
    from oracle import forward_ref
    config, bs, n_pts, h, w, n_cls = cfg
    frames = synth.make_batch(config, bs, n_points=n_pts, height=h, width=w)
    net = build(n_cls, n_pts, device)
    inputs = pyramid.frames_to_device(frames, device)
    taps, ref_taps = {}, {}
    with torch.no_grad():
        net(inputs, taps=taps)
        forward_ref.ffb6d_forward(dict(net.state_dict()), inputs, taps=ref_taps)
    assert sorted(taps) == sorted(ref_taps) and len(taps) == 14
    for k in sorted(taps):
        assert_close_scaled(taps[k].cpu().numpy(), ref_taps[k].cpu().numpy(), HOT_TOL, (cfg, k))


# bf16 bar (BASELINE.json configuration 5: bfloat16 activations / weights, fp32 accumulation): ~110 layers each rounding
# its output to 8 significant bits.  Bars = 2 x the error measured on the MI355X (profiles/r02_bf16_parity_vs_fp32_oracle.txt,
# relative to the tensor's range, against the fp32 plain-torch restatement): the three outputs max <= 2.4e-2 / mean <= 5.2e-3
# measured, the 14 fusion-stage embeddings max <= 1.4e-2 / mean <= 1.2e-3 measured; the test prints what it measures.
BF16_MAX, BF16_MEAN = 4.9e-2, 8e-3
BF16_STAGE_MAX, BF16_STAGE_MEAN = 2.8e-2, 2.4e-3


@pytest.mark.parametrize("cfg", [(7, 2, 1024, 120, 160, 5), (5, 2, 12288, 480, 640, 22)])
def test_bf16_forward_matches_fp32_oracle_at_bf16_tolerance(device, cfg):
    from oracle import forward_ref
    config, bs, n_pts, h, w, n_cls = cfg
    frames = synth.make_batch(config, bs, n_points=n_pts, height=h, width=w)
    net = build(n_cls, n_pts, device)
    net.precision = "bf16"
    inputs = pyramid.frames_to_device(frames, device)
    taps, ref_taps = {}, {}
    with torch.no_grad():
        ep = net(inputs, taps=taps)
        ref = forward_ref.ffb6d_forward(dict(net.state_dict()), inputs, taps=ref_taps)
    for k, want in list(ref.items()) + [(k, ref_taps[k]) for k in sorted(ref_taps)]:
        got = ep[k] if k in ep else taps[k]
        assert got.dtype == torch.float32 and got.shape == want.shape
        scale = float(want.abs().max())
        err = (got - want).abs()
        print(cfg, k, "bf16 max err / range %.3e  mean err / range %.3e" % (float(err.max()) / scale, float(err.mean()) / scale))
        bar_max, bar_mean = (BF16_MAX, BF16_MEAN) if k in ref else (BF16_STAGE_MAX, BF16_STAGE_MEAN)
        assert float(err.max()) <= bar_max * scale and float(err.mean()) <= bar_mean * scale, k
    net.precision = "fp32"                                   # and the same module answers in fp32 again (per-dtype weight caches)
    with torch.no_grad():
        ep32 = net(inputs)
    for k in ref:
        assert_close_scaled(ep32[k].cpu().numpy(), ref[k].cpu().numpy(), HOT_TOL, (cfg, k, "fp32 after bf16"))


def test_weight_updates_reach_the_fused_kernels(device):
    """Stale-cache guard (folded / padded / k-chunked / channels-last weights are cached per module): a model that already
    ran in eval() and then gets other weights -- load_state_dict, an in-place edit -- must answer like a fresh model."""
    frames = synth.make_batch(7, 1, n_points=1024, height=120, width=160)
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        shapes = json.load(fh)
    sd_a = synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=5)
    sd_b = synth.synth_state_dict_from_shapes(shapes, seed=1, n_classes=5)
    inputs = pyramid.frames_to_device(frames, device)
    net = M.FFB6D(n_classes=5, n_pts=1024)
    net.load_state_dict(sd_a)
    net = net.to(device).eval()
    fresh = M.FFB6D(n_classes=5, n_pts=1024)
    fresh.load_state_dict(sd_b)
    fresh = fresh.to(device).eval()
    with torch.no_grad():
        first = {k: v.clone() for k, v in net(inputs).items()}
        net.load_state_dict(sd_b)                      # same Parameter objects, new values
        second = net(inputs)
        want = fresh(inputs)
        for k in want:
            assert not torch.equal(first[k], second[k]), k
            assert_close_scaled(second[k].cpu().numpy(), want[k].cpu().numpy(), HOT_TOL, (k, "reload"))
        w = net.rndla_ds_stages[1].lfa.att_pooling_1.fc.weight
        w.mul_(1.5)                                    # in-place edit while staying in eval()
        fresh.rndla_ds_stages[1].lfa.att_pooling_1.fc.weight.mul_(1.5)
        third, want = net(inputs), fresh(inputs)
        for k in want:
            assert_close_scaled(third[k].cpu().numpy(), want[k].cpu().numpy(), HOT_TOL, (k, "edit"))
        assert any(not torch.equal(third[k], second[k]) for k in want)


def test_train_mode_without_grad_keeps_training_semantics(device):
    """train() under torch.no_grad() (BatchNorm re-calibration, a validation loop without eval()): the fused kernels
    fold RUNNING statistics, so they must not be used -- batch statistics, running-stat updates and the autograd-capable
    operators run instead, exactly like the unfused layers."""
    torch.manual_seed(0)
    mlp = M.SharedMLP(16, 8).to(device)
    x = torch.randn(2, 16, 300, 1, device=device)
    mlp.train()
    before = mlp.bn.bn.running_mean.clone()
    with torch.no_grad():
        got = mlp(x)
    assert not torch.equal(mlp.bn.bn.running_mean, before)          # statistics were updated
    y = torch.nn.functional.conv2d(x, mlp.conv.weight)
    mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
    want = torch.nn.functional.leaky_relu((y - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + 1e-6) *
                                          mlp.bn.bn.weight.view(1, -1, 1, 1) + mlp.bn.bn.bias.view(1, -1, 1, 1), 0.2)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    mlp.eval()
    with torch.no_grad():
        fused = mlp(x)                                             # eval + no_grad: folded running statistics
    assert not torch.allclose(fused, got)


def test_forward_small_matches_reference(device):
    gold = np.load(os.path.join(GOLDEN, "forward_small.npz"))
    frames = synth.make_batch(7, 2, n_points=1024, height=120, width=160)
    net = build(5, 1024, device)
    for idt in (torch.int64, torch.int32):
        inputs = pyramid.frames_to_device(frames, device, index_dtype=idt)
        with torch.no_grad():
            ep = net(inputs)
        for k in ("pred_rgbd_segs", "pred_kp_ofs", "pred_ctr_ofs"):
            got = ep[k].cpu().numpy()
            assert got.shape == gold[k].shape
            err = rel_err(got, gold[k])
            print(k, "max err / range vs the reference's CPU end_points", err)
            assert err <= E2E_TOL, (k, err)


def test_forward_full_size_matches_reference_sample(device):
    gold = np.load(os.path.join(GOLDEN, "forward_full_sample.npz"))
    frames = synth.make_batch(1, 1, n_points=12288)
    net = build(22, 12288, device)
    inputs = pyramid.frames_to_device(frames, device)
    with torch.no_grad():
        ep = net(inputs)
    assert ep["pred_rgbd_segs"].shape == (1, 22, 12288)
    assert ep["pred_kp_ofs"].shape == (1, 8, 12288, 3)
    assert ep["pred_ctr_ofs"].shape == (1, 1, 12288, 3)
    plain = oracle_on_device(net, inputs)
    for k in ("pred_rgbd_segs", "pred_kp_ofs", "pred_ctr_ofs"):
        scale = float(gold[k + "/absmax"])
        err = float(np.abs(ep[k].cpu().numpy().reshape(-1)[::97] - gold[k]).max()) / scale
        err_plain = float(np.abs(plain[k].cpu().numpy().reshape(-1)[::97] - gold[k]).max()) / scale
        print(k, "max rel err vs reference CPU: ours", err, "plain torch on GPU", err_plain)
        assert err <= E2E_TOL, (k, err)
        assert err <= 2 * err_plain + HOT_TOL   # we add nothing on top of the MIOpen-vs-CPU gap


def test_two_stream_forward_equals_single_stream(device):
    """The point branch runs on a second HIP stream under the colour branch's convolutions
    (forward_pm.forward); same kernels, same order per tensor -> same bits, unless
    MIOpen picked a split-K convolution that accumulates with atomics (then even two single-stream
    runs differ in the last bits and the comparison falls back to HOT_TOL of the output range).
    Repeated, with a NaN-filled block recycled through the allocator in between, so that a missing
    event / record_stream shows up as a race."""
    frames = synth.make_batch(3, 2, n_points=12288, height=480, width=640)
    net = build(22, 12288, device)
    inputs = pyramid.frames_to_device(frames, device)
    with torch.no_grad():
        net.two_streams = False
        want = {k: v.clone() for k, v in net(inputs).items()}
        again = net(inputs)
        reproducible = all(torch.equal(again[k], want[k]) for k in want)
        net.two_streams = True
        for rep in range(4):
            got = net(inputs)
            junk = torch.empty(64 << 20, device=device).fill_(float("nan"))    # recycle freed blocks
            del junk
            for k in want:
                if reproducible:
                    assert torch.equal(got[k], want[k]), (rep, k, float((got[k] - want[k]).abs().max()))
                else:
                    scale = float(want[k].abs().max())
                    assert float((got[k] - want[k]).abs().max()) <= HOT_TOL * scale, (rep, k)


@pytest.mark.parametrize("two_streams", [True, False])
def test_forward_builds_the_index_pyramid_itself_when_it_is_missing(device, two_streams):
    """Inputs that carry `dpt_xyz` instead of the 26 index tensors: the forward runs the dataset's 22 KNN searches
    (linemod_dataset.py:299-353) on the device -- the point-major path level by level on a third HIP stream under the
    network (forward_pm.StreamedPyramid) -- and must produce the bits of the forward fed with a prebuilt pyramid.
    Repeated with recycled NaN blocks so that a missing event / record_stream shows up as a race."""
    frames = synth.make_batch(9, 2, n_points=12288, height=480, width=640)
    net = build(22, 12288, device)
    net.two_streams = two_streams
    full = pyramid.frames_to_device(frames, device)
    lazy = {k: full[k] for k in ('rgb', 'cld_rgb_nrm', 'choose')}
    lazy['dpt_xyz'] = torch.from_numpy(frames['dpt_xyz']).to(device)
    with torch.no_grad():
        want = {k: v.clone() for k, v in net(full).items()}
        again = net(full)
        reproducible = all(torch.equal(again[k], want[k]) for k in want)
        for rep in range(3):
            got = net(lazy)
            junk = torch.empty(64 << 20, device=device).fill_(float("nan"))
            del junk
            for k in want:
                if reproducible:
                    assert torch.equal(got[k], want[k]), (rep, k, float((got[k] - want[k]).abs().max()))
                else:
                    assert float((got[k] - want[k]).abs().max()) <= HOT_TOL * float(want[k].abs().max()), (rep, k)
        with pytest.raises(KeyError):
            net({k: full[k] for k in ('rgb', 'cld_rgb_nrm', 'choose')})
    net.index_dtype = torch.int32
    with torch.no_grad():
        got = net(lazy)
    for k in want:
        assert float((got[k] - want[k]).abs().max()) <= HOT_TOL * float(want[k].abs().max()), k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_head_forms_equal_the_dense_last_stage(device, precision, n_pts=12288, height=480, width=640, n_frames=2):
    """forward_pm's forms around the prediction heads -- the last colour stage evaluated at the picked pixels only
    (LAST_STAGE_AT_CHOSEN), first layers as one stacked GEMM, the remaining layers of a head as one launch, keypoint head on the side stream -- against the
    same network with all of them off (the full 480 x 640 map, then the `choose` pick; ffb6d.py:302-318).  fp32: the hot-path bar
    (the K = 576 GEMM sums the convolution's products in another order than the dense path); bf16: the bf16 bar of the whole-forward parity tests (5e-2 of the range)."""
    from ffb6d_amd import forward_pm
    frames = synth.make_batch(11, n_frames, n_points=n_pts, height=height, width=width)
    net = build(22, n_pts, device)
    net.precision = precision
    inputs = pyramid.frames_to_device(frames, device)
    names = ("LAST_STAGE_AT_CHOSEN", "HEADS_SHARE_FIRST", "HEADS_ALIGN_LAST", "HEADS_ON_BOTH_STREAMS", "HEADS_CHAIN_FUSED")
    keep = {n: getattr(forward_pm, n) for n in names}
    chain = forward_pm.ops_pm.mlp_chain3
    on_gpu = torch.device(device).type == "cuda"          # the emulator suite calls this with CPU tensors: one stream, the fused path directly
    run = (lambda: net(inputs)) if on_gpu else (lambda: forward_pm.forward(net, inputs, {}, two_streams=False))
    try:
        with torch.no_grad():
            for n in names:
                setattr(forward_pm, n, False)
            want = {k: v.float().clone() for k, v in run().items()}
            for n in names:
                setattr(forward_pm, n, True)
            calls, chain = [], forward_pm.ops_pm.mlp_chain3
            forward_pm.ops_pm.mlp_chain3 = lambda *a, **k: (calls.append(1), chain(*a, **k))[1]
            got = {k: v.float() for k, v in run().items()}
            assert len(calls) == 3                                           # the fused chain: one launch per head, fp32 and (round 5) bf16
    finally:
        forward_pm.ops_pm.mlp_chain3 = chain
        for n, v in keep.items():
            setattr(forward_pm, n, v)
    for k in want:
        scale = float(want[k].abs().max())
        err = float((got[k] - want[k]).abs().max()) / scale
        print(precision, k, "max err / range", err)
        assert err <= (HOT_TOL if precision == "fp32" else 5e-2), (k, err)


def test_batch_items_are_independent(device):
    """Every op on the path is per-sample in eval mode: a frame's result must not depend on its batch
    neighbours -- the property multi-GPU sharding relies on.  Two checks: (1) the same frame between DIFFERENT neighbours in batches
    of the same size (same MIOpen algorithms, cudnn.benchmark off: whatever differs comes from the batch neighbours) -- within the
    hot-path bar, 1e-5 of the output range (measured 1.6e-6 on the logits: MIOpen kernels that accumulate across the batch dimension
    with atomics; the hand-written kernels are per frame by construction, tests/test_pm_gpu.py); (2) the frame alone
    against the frame in a batch of three, where MIOpen may pick another algorithm per batch size: 1e-3 of range as before."""
    keep = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    try:
        frames = synth.make_batch(7, 5, n_points=1024, height=120, width=160)
        net = build(5, 1024, device)
        inputs = pyramid.frames_to_device(frames, device)
        pick = lambda idx: {k: v[idx].contiguous() for k, v in inputs.items()}       # noqa: E731
        with torch.no_grad():
            a = net(pick([0, 1, 2]))
            b = net(pick([3, 1, 4]))
            single = net(pick([1]))
        for k in a:
            scale = float(a[k].abs().max())
            err = float((a[k][1] - b[k][1]).abs().max()) / scale
            print(k, "same frame, other neighbours: max err / range", err)
            assert err <= HOT_TOL, (k, err)
            assert float((a[k][1:2] - single[k]).abs().max()) <= 1e-3 * scale, k
    finally:
        torch.backends.cudnn.benchmark = keep


def _step_gradients(device, n_pts, height, width, batch, autocast):
    """gradients of every parameter for one training forward / backward: ours (stock modules + the row operators of ops_cl, backward
    through csrc/train_rows.hip / train_ops.hip), plain torch autograd of oracle/forward_ref.py in fp32, and -- with autocast -- plain
    torch under the same bf16 autocast"""
    from oracle import forward_ref
    frames = synth.make_batch(7, batch, n_points=n_pts, height=height, width=width)
    net = build(5, n_pts, device)         # eval(): BatchNorm uses running statistics in both paths
    inputs = pyramid.frames_to_device(frames, device)
    params = dict(net.named_parameters())

    def loss_of(ep):
        return sum((v.float() ** 2).mean() for v in ep.values())

    def ref_run(cast):
        # one leaf per TENSOR: `final` is shared by two decoder stages (ffb6d.py:86-87), its weight appears under two state_dict keys
        leaves, sd = {}, {}
        for k, v in net.state_dict().items():
            key = (v.data_ptr(), tuple(v.shape))
            if key not in leaves:
                leaves[key] = v.detach().clone().requires_grad_(k in params)
            sd[k] = leaves[key]
        with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=cast):
            ref_loss = loss_of(forward_ref.ffb6d_forward(sd, inputs))
        ref_loss.backward()
        return float(ref_loss), {k: sd[k].grad for k in params}

    net.zero_grad()
    with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        loss = loss_of(net(inputs))
    loss.backward()
    ours = {n: p.grad.detach().clone() if p.grad is not None else None for n, p in params.items()}
    ref_loss, ref = ref_run(False)
    plain = ref_run(True) if autocast else None
    return float(loss), ours, ref_loss, ref, plain


@pytest.mark.parametrize("n_pts,height,width,batch", [(1024, 120, 160, 2), (1100, 136, 168, 2),      # the second: ragged against every tile size
                                                      (12288, 480, 640, 1)])                         # the benchmarked geometry
def test_training_step_gradients_match_plain_torch(device, n_pts, height, width, batch):
    """Config-3 path (DDP training) on one rank: gradients through the custom operators' backward
    kernels (inverted-index row sums / arg-max / softmax backward) inside the whole network must equal the
    gradients plain torch autograd produces for the same forward (oracle/forward_ref.py) -- for EVERY parameter."""
    loss, ours, ref_loss, ref, _ = _step_gradients(device, n_pts, height, width, batch, False)
    assert abs(loss - ref_loss) <= 1e-3 * abs(ref_loss)
    worst, checked, loose = ("", 0.0), 0, 0
    for n, r in ref.items():
        g = ours[n]
        if r is None:                                  # a parameter the forward does not reach (none today)
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert g is not None, n
        scale = float(r.abs().max())
        if scale == 0.0:
            assert float(g.abs().max()) == 0.0, n
            continue
        err = float((g - r).abs().max()) / scale
        checked += 1
        if err > worst[1]:
            worst = (n, err)
        # measured 5e-6 .. 3e-4 on most parameters, up to 7e-3 on a handful run to run: MIOpen's backward-data/weight algorithms and the
        # float atomics of the scatter-add kernels are not bit-reproducible (the largest element of a gradient is the yardstick)
        assert err <= 1e-2, (n, err)
        loose += err > 5e-3
    print("parameters checked:", checked, "worst:", worst, "above 5e-3:", loose)
    assert checked >= 300
    assert loose <= max(3, checked // 50), loose          # the round-4 bar (5e-3) still holds for all but a handful of parameters per run


def test_training_step_gradients_under_bf16_autocast(device):
    """the same step as the reference trains it (apex amp ~ torch.autocast(bfloat16), train_lm.py:600): our gradients may be as far from
    the fp32 gradients as plain torch's gradients under the same autocast are -- twice that plus a floor, per parameter, measured in the
    Frobenius norm (bf16 rounding is noise on every element; the largest element is not a stable yardstick)."""
    loss, ours, ref_loss, ref, (plain_loss, plain) = _step_gradients(device, 1100, 136, 168, 2, True)
    # (the objective itself -- sum of the mean squares of the three outputs -- moves by ~10 % under bf16 autocast on these weights, ours as plain torch's)
    assert abs(loss - ref_loss) <= 2.0 * abs(plain_loss - ref_loss) + 1e-2 * abs(ref_loss), (loss, plain_loss, ref_loss)
    worst, checked = ("", 0.0, 0.0), 0
    for n, r in ref.items():
        if r is None or float(r.abs().max()) == 0.0:
            continue
        nr = float(r.double().norm())
        e_ours = float((ours[n].double() - r.double()).norm()) / nr
        e_plain = float((plain[n].double() - r.double()).norm()) / nr
        checked += 1
        if e_ours - 2 * e_plain > worst[1] - 2 * worst[2]:
            worst = (n, e_ours, e_plain)
        assert e_ours <= 2.0 * e_plain + 2e-2, (n, e_ours, e_plain)
    print("parameters checked:", checked, "worst (name, ours, plain torch):", worst)
    assert checked >= 300
