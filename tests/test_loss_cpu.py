"""ffb6d_amd/loss.py (the training objective bench.py --mode train times) against the reference's own loss module
(ffb6d/models/loss.py, imported from /root/reference when present) and against hand-computed values; synthetic targets."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from ffb6d_amd import loss, synth

REF_LOSS = "/root/reference/ffb6d/models/loss.py"


def case(seed, B=2, C=5, N=64, K=8):
    g = torch.Generator().manual_seed(seed)
    ep = {"pred_rgbd_segs": torch.randn(B, C, N, generator=g), "pred_kp_ofs": torch.randn(B, K, N, 3, generator=g),
          "pred_ctr_ofs": torch.randn(B, 1, N, 3, generator=g)}
    labels = torch.randint(0, C, (B, N), generator=g)
    labels[0, : N // 2] = 0                                  # background points carry no offset loss
    return ep, labels, torch.randn(B, N, K, 3, generator=g), torch.randn(B, N, 1, 3, generator=g)


@pytest.mark.reference
@pytest.mark.parametrize("seed", [0, 1])
def test_objective_equals_the_reference_modules(seed):
    spec = importlib.util.spec_from_file_location("ref_loss", REF_LOSS)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ep, labels, kp_t, ctr_t = case(seed)
    outs = []
    for ours in (False, True):
        leaves = {k: v.clone().requires_grad_(True) for k, v in ep.items()}
        if ours:
            total, terms = loss.training_loss(leaves, labels, kp_t, ctr_t)
            terms = [terms["loss_rgbd_seg"], terms["loss_kp_of"], terms["loss_ctr_of"]]
        else:                                                # train_lm.py:245-259 with criterion = FocalLoss(gamma=2), OFLoss()
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")              # F.log_softmax without dim (loss.py:31)
                seg = ref.FocalLoss(gamma=2)(leaves["pred_rgbd_segs"], labels.view(-1)).sum()
            kp = ref.OFLoss()(leaves["pred_kp_ofs"], kp_t, labels).sum()
            ctr = ref.OFLoss()(leaves["pred_ctr_ofs"], ctr_t, labels).sum()
            total, terms = 2.0 * seg + kp + ctr, [seg, kp, ctr]
        total.backward()
        outs.append(([float(t) for t in terms] + [float(total)], [leaves[k].grad for k in sorted(leaves)]))
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6, atol=1e-7)
    for a, b in zip(outs[0][1], outs[1][1]):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-7)


def test_focal_term_by_hand_and_its_constant_modulating_factor():
    logits = torch.tensor([[2.0, 0.0, -1.0], [0.5, 0.5, 0.5]], requires_grad=True)
    labels = torch.tensor([0, 2])
    p = torch.softmax(logits.detach(), dim=1)
    pt = torch.stack([p[0, 0], p[1, 2]])
    want = (-((1 - pt) ** 2) * pt.log()).mean()
    got = loss.focal_loss(logits, labels)
    assert abs(float(got) - float(want)) < 1e-7
    got.backward()
    # gradient of -(1-pt)^2 log pt with (1-pt)^2 held constant: (1-pt)^2 (softmax - onehot) / M
    onehot = torch.zeros(2, 3).scatter_(1, labels.view(-1, 1), 1.0)
    torch.testing.assert_close(logits.grad, ((1 - pt) ** 2).view(-1, 1) * (p - onehot) / 2, rtol=1e-5, atol=1e-7)


def test_offset_term_counts_object_points_only():
    pred = torch.zeros(1, 2, 4, 3)
    targ = torch.ones(1, 4, 2, 3)
    labels = torch.tensor([[0, 3, 3, 0]])
    got = loss.offset_l1_loss(pred, targ, labels)
    assert got.shape == (1, 2)
    np.testing.assert_allclose(got.numpy(), np.full((1, 2), 6.0 / (2 + 1e-3)), rtol=1e-6)       # 2 points x 3 coordinates x |1|
    assert float(loss.offset_l1_loss(pred, targ, torch.zeros(1, 4, dtype=torch.long)).abs().max()) == 0.0


def test_synthetic_targets_have_the_dataset_shapes_and_leave_the_frames_alone():
    a = synth.make_frame(4242, n_points=512, height=60, width=80)
    t = synth.make_targets(4242, a["cld"], n_classes=22, n_kps=8)
    b = synth.make_frame(4242, n_points=512, height=60, width=80)
    assert all(np.array_equal(a[k], b[k]) for k in a)                                               # own random stream
    assert t["labels"].shape == (512,) and t["kp_targ_ofst"].shape == (512, 8, 3) and t["ctr_targ_ofst"].shape == (512, 1, 3)
    fg = t["labels"] > 0
    assert 0 < fg.sum() < 512 and t["labels"].max() < 22
    assert not t["kp_targ_ofst"][~fg].any() and not t["ctr_targ_ofst"][~fg].any()
    ctr = a["cld"][fg] + t["ctr_targ_ofst"][fg, 0]                                                  # point + offset = its object's centre
    for c in np.unique(t["labels"][fg]):
        pts = ctr[t["labels"][fg] == c]
        assert np.abs(pts - pts[0]).max() < 1e-5 or len(np.unique(np.round(pts, 4), axis=0)) <= 3   # (two spheres may share a class)
    t2 = synth.make_targets(4242, a["cld"], n_classes=22, n_kps=8)
    assert all(np.array_equal(t[k], t2[k]) for k in t)
