"""ops_cl.batch_norm_act (train-mode BatchNorm + activation of a shared MLP on rows; opt-in in the model: FFB6D_BN_ROWS=1) on the
SIMT emulator against torch's own BatchNorm + activation: output, running statistics, and the gradients of input, weight and
bias -- batch statistics and running statistics, fp32 and bf16 rows, channel counts with and without a power-of-two unit count."""
import pytest
import torch

from ffb6d_amd import ops_cl

F = torch.nn.functional


def reference(x, bn, act, slope):
    y = bn(x)
    return y if act == 0 else (torch.relu(y) if act == 1 else F.leaky_relu(y, slope))


def run(fn, x, bn):
    bn.zero_grad()
    xs = x.detach().clone().requires_grad_(True)
    y = fn(xs, bn)
    r = torch.rand(y.shape, generator=torch.Generator().manual_seed(5)).to(y.dtype)
    (y.float() * r.float()).sum().backward()
    return [y.detach().float(), xs.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,act", [((2, 16, 50, 16), 2), ((1, 64, 37, 1), 1), ((3, 24, 5, 7), 0), ((2, 8, 300, 1), 2), ((1, 256, 9, 2), 1)])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_act_rows_matches_torch(emu, dt, shape, act, training):
    g = torch.Generator().manual_seed(sum(shape) + act)
    C = shape[1]
    x = (2.0 * torch.randn(*shape, generator=g) + 3.0).contiguous(memory_format=torch.channels_last)        # mean well away from 0
    outs = []
    for ours in (False, True):
        torch.manual_seed(1)
        bn = torch.nn.BatchNorm2d(C, eps=1e-6, momentum=0.99)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
            bn.running_mean.normal_()
            bn.running_var.uniform_(0.5, 2.0)
        bn.train(training)
        if ours:
            outs.append(run(lambda t, m: ops_cl.batch_norm_act(t.to(dt), m, act, 0.2), x, bn))
        else:                         # torch in fp32 on the values the kernel sees
            outs.append(run(lambda t, m: reference(t.to(dt).float(), m, act, 0.2), x, bn))
        assert int(bn.num_batches_tracked) == (1 if training else 0)
    bar = 2e-5 if dt == torch.float32 else 1.5e-2
    for name, a, b in zip(("y", "gx", "gw", "gb", "running_mean", "running_var"), *outs):
        err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)
        assert err <= bar, (name, err)


def test_unsupported_cases_go_to_torch(emu):
    bn = torch.nn.BatchNorm2d(6).train()
    x = torch.randn(2, 6, 5, 5)
    torch.testing.assert_close(ops_cl.batch_norm_act(x, bn, 1), torch.relu(torch.nn.BatchNorm2d(6).train()(x)))


def test_training_step_gradients_with_the_row_batch_norm(emu, monkeypatch):
    """the whole-network gradient check (oracle/forward_ref.py's plain-torch forward as the reference) with FFB6D_BN_ROWS=1: every
    shared MLP of both branches normalises and activates through ops_cl.batch_norm_act"""
    import test_forward_gpu as TF
    monkeypatch.setenv("FFB6D_BN_ROWS", "1")
    calls = []
    real = ops_cl.batch_norm_act
    monkeypatch.setattr(ops_cl, "batch_norm_act", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    TF.test_training_step_gradients_match_plain_torch(torch.device("cpu"))
    assert len(calls) >= 55                      # the 60 two-dimensional shared MLPs with BatchNorm of one forward
