"""CPU suite: the multi-process harness of bench.py on the gloo backend, world_size 2
(the GPU runs use the same code on nccl = RCCL)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT


def run_ranks(script, world, tmp_path, timeout, **extra_env):
    """one process per rank; every rank's output goes to its own FILE (a pipe that the parent reads rank by rank fills up when a
    later rank is chatty -- e.g. rebuilds the emulated library -- and the ranks then deadlock in their first collective)"""
    port = free_port()
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   **extra_env)
        log = open(tmp_path / ("rank%d.log" % rank), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=log, stderr=subprocess.STDOUT, text=True))
    try:
        for p in procs:
            p.wait(timeout=timeout)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    outs = []
    for log in logs:
        log.seek(0)
        outs.append(log.read())
        log.close()
    assert all(p.returncode == 0 for p in procs), [o[-3000:] for o in outs]
    return outs


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from ffb6d_amd import distributed as D, synth
    g = D.init_from_env(backend="gloo")
    frames = D.shard_frames(3, 2, g.rank, None, n_points=256, height=60, width=80)
    calls = []
    def step(timed):
        calls.append(timed)
        time.sleep(0.02 * (1 + g.rank))          # rank 1 is the slow one
    elapsed = D.timed_steps(step, warmup=2, steps=3, group=g)
    total = g.sum_over_ranks(float(frames["cld"].sum()))
    # DDP gradient all-reduce (BASELINE config 3 recipe, train_lm.py:625-628) through wrap_ddp
    import torch
    torch.manual_seed(0)
    lin = torch.nn.Linear(3, 2)
    ddp = D.wrap_ddp(lin, torch.device("cpu"))
    x = torch.from_numpy(frames["cld"][0][:16])
    ddp(x).pow(2).mean().backward()
    grad = lin.weight.grad.flatten().tolist()
    local = torch.nn.Linear(3, 2)
    local.load_state_dict({k: v.clone() for k, v in lin.state_dict().items()})
    local(x).pow(2).mean().backward()
    # the collective counter the training bench and the RCCL tests use (bench.py train_extra, tests/test_multigpu_gpu.py)
    import torch.distributed as dist
    with D.count_collectives() as cc:
        t = torch.ones(3)
        dist.all_reduce(t)
        dist.all_reduce(t)
        dist.broadcast(t, src=0)
    restored = dist.all_reduce.__module__.startswith("torch.distributed")
    out = dict(rank=g.rank, world=g.world, elapsed=elapsed, calls=calls,
               first_seed_check=float(frames["cld"][0].sum()), total=total, grad=grad,
               local_grad=local.weight.grad.flatten().tolist(), counted=cc.counts, restored=restored,
               buckets=D.ddp_bucket_count(ddp))
    print("RESULT " + json.dumps(out), flush=True)
    g.close()
""") % ROOT


def test_two_rank_gloo_harness(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outs = run_ranks(script, 2, tmp_path, 120)
    res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs]
    res.sort(key=lambda r: r["rank"])
    assert [r["world"] for r in res] == [2, 2]
    # exactly W untimed + K timed steps on every rank
    assert all(r["calls"] == [False, False, True, True, True] for r in res)
    # both ranks report the SAME elapsed = the slow rank's (max over ranks): 3 * 40 ms
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9
    assert res[0]["elapsed"] >= 3 * 0.04 * 0.9
    # disjoint shards: rank r holds samples [2r, 2r+1] of config 3
    from ffb6d_amd import synth
    for r in res:
        want = synth.make_frame(synth.frame_seed(3, 2 * r["rank"]), n_points=256, height=60, width=80)
        assert abs(r["first_seed_check"] - float(want["cld"].sum())) < 1e-3
    assert abs(res[0]["total"] - res[1]["total"]) < 1e-6   # SUM all-reduce agrees
    # DDP: both ranks end up with the SAME gradient = mean of the two local gradients
    np.testing.assert_allclose(res[0]["grad"], res[1]["grad"], rtol=1e-6, atol=1e-7)
    mean = (np.array(res[0]["local_grad"]) + np.array(res[1]["local_grad"])) / 2
    np.testing.assert_allclose(res[0]["grad"], mean, rtol=1e-5, atol=1e-6)
    assert not np.allclose(res[0]["local_grad"], res[1]["local_grad"])   # ranks really saw different frames
    assert all(r["counted"] == {"all_reduce": 2, "broadcast": 1} and r["restored"] for r in res)
    assert all(r["buckets"] in (None, 1) for r in res)      # one small bucket (or no logging data in this torch build)


def test_single_process_group_is_a_noop():
    from ffb6d_amd import distributed as D
    g = D.Group()
    g.barrier()
    assert g.max_over_ranks(1.5) == 1.5 and g.sum_over_ranks(2.0) == 2.0
    n = []
    t = D.timed_steps(lambda timed: n.append(timed), warmup=1, steps=2, group=g)
    assert n == [False, True, True] and t >= 0


EIGHT_WORKER = textwrap.dedent("""
    import hashlib, json, os, sys
    sys.path.insert(0, %r)
    import torch
    from ffb6d_amd import distributed as D
    g = D.init_from_env(backend="gloo")
    frames = D.shard_frames(3, 2, g.rank, None, n_points=256, height=60, width=80)
    digests = [hashlib.sha256(frames["cld"][i].tobytes() + frames["rgb"][i].tobytes()).hexdigest() for i in range(2)]
    t = D.timed_steps(lambda timed: None, warmup=1, steps=2, group=g)
    total = g.sum_over_ranks(float(g.rank))
    slowest = g.max_over_ranks(float(g.rank))
    print("RESULT " + json.dumps(dict(rank=g.rank, world=g.world, digests=digests, total=total, slowest=slowest, t=t)))
    g.close()
""") % ROOT


def test_eight_rank_sharding_is_disjoint_and_covers_the_global_batch(tmp_path):
    """BASELINE configuration 3's launch shape on CPU: eight gloo ranks, two frames each -- every rank holds frames
    [2r, 2r + 2) of the synthetic stream (all sixteen different, the same frames a single process would generate for those seeds),
    the control collectives (barrier, MAX / SUM over ranks) see all eight ranks"""
    from ffb6d_amd import synth
    script = tmp_path / "eight_worker.py"
    script.write_text(EIGHT_WORKER)
    outs = run_ranks(script, 8, tmp_path, 600, OMP_NUM_THREADS="1")
    res = sorted((json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs), key=lambda r: r["rank"])
    assert [r["rank"] for r in res] == list(range(8)) and all(r["world"] == 8 for r in res)
    assert all(r["total"] == 28.0 and r["slowest"] == 7.0 for r in res)
    digests = [d for r in res for d in r["digests"]]
    assert len(set(digests)) == 16
    import hashlib
    for s in (0, 5, 15):            # rank s // 2 holds sample s of the stream
        f = synth.make_frame(synth.frame_seed(3, s), n_points=256, height=60, width=80)
        assert hashlib.sha256(f["cld"].tobytes() + f["rgb"].tobytes()).hexdigest() == digests[s]


TRAIN_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import pytest, torch
    from tests.simt import bind
    mp = pytest.MonkeyPatch()
    bind.bind(mp)                       # CPU tensors through the emulated kernels: the real training forward / backward
    from ffb6d_amd import distributed as D, loss, model as M, pyramid, synth
    g = D.init_from_env(backend="gloo")
    frames = D.shard_frames(3, 1, g.rank, None, n_points=1024, height=120, width=160)
    tg = synth.make_targets(synth.frame_seed(3, g.rank), frames["cld"][0], n_classes=4)
    targets = (torch.from_numpy(tg["labels"])[None].long(), torch.from_numpy(tg["kp_targ_ofst"])[None], torch.from_numpy(tg["ctr_targ_ofst"])[None])
    torch.manual_seed(0)                                    # same initial weights on both ranks
    net = M.FFB6D(n_classes=4, n_pts=1024).train()
    inputs = pyramid.frames_to_device(frames, torch.device("cpu"))
    names = ["rndla_ds_stages.0.lfa.att_pooling_1.fc.weight", "cnn_ds_stages.3.0.stages.1.1.weight", "kp_ofst_layer.3.conv.weight",
             "ds_fuse_p2r_pre_layers.0.conv.weight", "rndla_up_stages.2.conv.weight"]
    def grads_of(module):
        module.zero_grad()
        torch.manual_seed(7)                                # same dropout masks in both passes
        out = module(inputs)
        loss.training_loss(out, *targets)[0].backward()      # the reference's objective (train_lm.py:245-259) on synthetic targets
        params = dict(net.named_parameters())
        return {n: params[n].grad.flatten()[:64].tolist() for n in names}
    local = grads_of(net)
    ddp = D.wrap_ddp(net, torch.device("cpu"))              # gloo: BatchNorm statistics stay per rank
    synced = grads_of(ddp)
    print("RESULT " + json.dumps(dict(rank=g.rank, local=local, synced=synced)), flush=True)
    g.close()
""") % ROOT


def test_two_rank_training_step_averages_gradients_through_the_custom_operators(tmp_path):
    """BASELINE config 3 on two gloo ranks, one frame each, the REAL network: forward through the stock modules + the channels-last
    neighbour operators (ops_cl: row gathers with a shared inverted index, max-pool, attentive pooling, row log-softmax, folded
    pyramid pooling), backward through their kernels on the SIMT emulator, DistributedDataParallel around it (train_lm.py:625-628).
    Every rank must end with the mean of the two ranks' local gradients -- i.e. every parameter took part in the all-reduce and the
    custom autograd Functions (non-tensor arguments, shared plans) are DDP-clean."""
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    outs = run_ranks(script, 2, tmp_path, 900, OMP_NUM_THREADS="4")
    res = sorted((json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs), key=lambda r: r["rank"])
    for name in res[0]["local"]:
        l0, l1 = np.array(res[0]["local"][name]), np.array(res[1]["local"][name])
        s0, s1 = np.array(res[0]["synced"][name]), np.array(res[1]["synced"][name])
        scale = max(np.abs(l0).max(), np.abs(l1).max())
        assert scale > 0 and not np.allclose(l0, l1, rtol=1e-3, atol=1e-6 * scale), name          # the ranks saw different frames
        np.testing.assert_allclose(s0, s1, rtol=0, atol=1e-6 * scale, err_msg=name)                  # one gradient on both ranks
        np.testing.assert_allclose(s0, (l0 + l1) / 2, rtol=0, atol=2e-5 * scale, err_msg=name)       # = the mean of the local ones
