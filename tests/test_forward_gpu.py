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
E2E_TOL = 1e-2


def build(n_classes, n_pts, device):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        shapes = json.load(fh)
    net = M.FFB6D(n_classes=n_classes, n_pts=n_pts)
    net.load_state_dict(synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=n_classes))
    return net.to(device).eval()


def rel_err(got, want):
    scale = max(float(np.abs(want).max()), 1.0)
    return float(np.abs(got - want).max()) / scale


def oracle_on_device(net, inputs):
    from oracle import forward_ref
    sd = {k: v for k, v in net.state_dict().items()}
    with torch.no_grad():
        return forward_ref.ffb6d_forward(sd, {k: (v.long() if v.dtype == torch.int32 else v) for k, v in inputs.items()})


@pytest.mark.parametrize("cfg", [(7, 2, 1024, 120, 160, 5), (1, 1, 12288, 480, 640, 22), (4, 1, 24576, 480, 640, 22)])
def test_hot_path_matches_plain_torch_on_the_same_device(device, cfg):
    config, bs, n_pts, h, w, n_cls = cfg
    frames = synth.make_batch(config, bs, n_points=n_pts, height=h, width=w)
    net = build(n_cls, n_pts, device)
    inputs = pyramid.frames_to_device(frames, device)
    with torch.no_grad():
        ep = net(inputs)
    ref = oracle_on_device(net, inputs)
    for k in ("pred_rgbd_segs", "pred_kp_ofs", "pred_ctr_ofs"):
        err = rel_err(ep[k].cpu().numpy(), ref[k].cpu().numpy())
        print(cfg, k, "hot-path max rel err", err)
        assert err <= HOT_TOL, (k, err)


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
            print(k, "max rel err", err)
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
        scale = max(float(gold[k + "/absmax"]), 1.0)
        err = float(np.abs(ep[k].cpu().numpy().reshape(-1)[::97] - gold[k]).max()) / scale
        err_plain = float(np.abs(plain[k].cpu().numpy().reshape(-1)[::97] - gold[k]).max()) / scale
        print(k, "max rel err vs reference CPU: ours", err, "plain torch on GPU", err_plain)
        assert err <= E2E_TOL, (k, err)
        assert err <= 2 * err_plain + HOT_TOL   # we add nothing on top of the MIOpen-vs-CPU gap


def test_two_stream_forward_equals_single_stream(device):
    """The point branch runs on a second HIP stream under the colour branch's convolutions
    (model.FFB6D._forward_two_streams); same kernels, same order per tensor -> same bits, unless
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
                    scale = max(float(want[k].abs().max()), 1.0)
                    assert float((got[k] - want[k]).abs().max()) <= HOT_TOL * scale, (rep, k)


def test_batch_items_are_independent(device):
    """Every op on the path is per-sample in eval mode: a frame's
    result must not depend on its batch neighbours -- the property multi-GPU sharding relies on."""
    frames = synth.make_batch(7, 3, n_points=1024, height=120, width=160)
    net = build(5, 1024, device)
    inputs = pyramid.frames_to_device(frames, device)
    one = {k: v[1:2].contiguous() for k, v in inputs.items()}
    with torch.no_grad():
        full = net(inputs)
        single = net(one)
    for k in full:   # MIOpen may pick another algorithm per batch size: compare at range scale
        scale = float(full[k].abs().max())
        assert float((full[k][1:2] - single[k]).abs().max()) <= 1e-3 * scale, k


def test_training_step_gradients_match_plain_torch(device):
    """Config-3 path (DDP training) on one rank: gradients through the custom operators' backward
    kernels (scatter-add / arg-max / softmax backward) inside the whole network must equal the
    gradients plain torch autograd produces for the same forward (oracle/forward_ref.py)."""
    from oracle import forward_ref
    frames = synth.make_batch(7, 2, n_points=1024, height=120, width=160)
    net = build(5, 1024, device)          # eval(): BatchNorm uses running statistics in both paths
    inputs = pyramid.frames_to_device(frames, device)
    names = ["rndla_ds_stages.0.lfa.mlp1.conv.weight", "rndla_ds_stages.2.lfa.att_pooling_1.fc.weight",
             "ds_fuse_r2p_pre_layers.1.conv.weight", "ds_fuse_p2r_pre_layers.0.conv.weight",
             "rndla_up_stages.1.conv.weight", "cnn_ds_stages.0.0.conv1.weight", "rndla_pre_stages.conv.weight"]
    params = dict(net.named_parameters())

    def loss_of(ep):
        return sum((v.float() ** 2).mean() for v in ep.values())

    net.zero_grad()
    with torch.enable_grad():
        loss = loss_of(net(inputs))
    loss.backward()
    ours = {n: params[n].grad.detach().clone() for n in names}

    sd = {k: v.detach().clone().requires_grad_(k in params) for k, v in net.state_dict().items()}
    with torch.enable_grad():
        ref_loss = loss_of(forward_ref.ffb6d_forward(sd, inputs))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss))
    for n in names:
        g, r = ours[n], sd[n].grad
        scale = float(r.abs().max())
        assert scale > 0, n
        err = float((g - r).abs().max()) / scale
        print(n, "grad max rel err", err)
        # measured 5e-6 .. 3e-4 run to run: MIOpen's backward-data/weight algorithms and the float
        # atomics of the scatter-add kernels are not bit-reproducible
        assert err <= 5e-3, (n, err)
