"""The RCCL path of BASELINE configuration 3 (train_lm.py:560-563,592,625-628: NCCL process group -> SyncBatchNorm -> DistributedDataParallel)
on real devices: skipped unless the box has at least two GPUs (the round's own GPU box has one; the driver's 8-GPU node runs them).
Until round 5 every multi-rank rehearsal ran over gloo, where BatchNorm statistics stay per rank by construction: `backend="nccl"`,
`device_id=`, SyncBatchNorm over channels-last bf16 rows and the gradient all-reduce over xGMI had executed on no device.

  * bench.py --gpus 2 (inference, and --mode train in bf16 with SyncBatchNorm) through the driver's own launch path;
  * two RCCL ranks, two frames each: the gradients DistributedDataParallel leaves on every rank equal the gradients of ONE process on
    the four frames (SyncBatchNorm makes the forward the four-frame forward; the all-reduce averages the ranks' gradients).
The second check's worker also runs as a ONE-rank rehearsal on any GPU box (same script, no collective), so the script itself is exercised
every round."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _bench(*flags, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@two_gpus
def test_bench_two_gpus_over_rccl_inference():
    line = _bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert "nccl" in line["config"]["parallelism"]
    assert abs(line["value"] - 16 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


@two_gpus
def test_bench_two_gpus_over_rccl_training_with_sync_batchnorm_in_bf16():
    line = _bench("--gpus", "2", "--mode", "train", "--precision", "bf16", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                  "--cudnn-benchmark", "0", timeout=1200)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16
    assert "train" in line["metric"] and "SyncBatchNorm" in line["config"]["workload"] and line["value"] > 0
    # one invocation prices SyncBatchNorm: collectives of a step counted, the same steps timed with per-rank statistics as well
    extra = line["train_extra"]
    print("train_extra:", json.dumps(extra))
    n_bn = extra["sync_batchnorm_layers"]
    calls = sum(extra["python_side_collectives_per_step"].values())
    assert n_bn > 50 and calls >= 2 * n_bn                       # one gather of the statistics forward, one reduction backward, per layer
    assert set(extra["bn_variants"]) == {"sync_batchnorm", "local_batchnorm"} and extra["bn_variants"]["local_batchnorm"]["frames_per_s"] > 0


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import torch.distributed as dist
    from ffb6d_amd import distributed as D, loss, model as M, pyramid, synth
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    g = D.init_from_env(backend="nccl", device=dev)
    PER, NPT, H, W, NCLS = 2, 1024, 120, 160, 4

    def batch(first, count):
        fr = [synth.make_frame(synth.frame_seed(3, first + s), n_points=NPT, height=H, width=W) for s in range(count)]
        frames = {k: np.stack([f[k] for f in fr]) for k in fr[0]}
        tg = [synth.make_targets(synth.frame_seed(3, first + s), frames["cld"][s], n_classes=NCLS) for s in range(count)]
        t = tuple(torch.from_numpy(np.stack([x[k] for x in tg])).to(dev) for k in ("labels", "kp_targ_ofst", "ctr_targ_ofst"))
        inputs = pyramid.frames_to_device(frames, dev)
        inputs["rgb"] = inputs["rgb"].contiguous(memory_format=torch.channels_last)
        return inputs, (t[0].long(),) + t[1:]

    def build():
        net = M.FFB6D(n_classes=NCLS, n_pts=NPT)              # the same weights everywhere: the seeded synthetic state dict of the forward tests
        with open(os.path.join(%r, "tests", "golden", "state_dict_keys.json")) as fh:
            net.load_state_dict(synth.synth_state_dict_from_shapes(json.load(fh), seed=0, n_classes=NCLS))
        net = net.to(dev).to(memory_format=torch.channels_last).train()
        for m in net.modules():
            if isinstance(m, torch.nn.modules.dropout._DropoutNd):
                m.p = 0.0                                     # (Dropout and Dropout2d) masks would depend on the batch split
        return net

    def objective(out, targets, offsets_scale):
        total, terms = loss.training_loss(out, *targets)
        # DDP AVERAGES the ranks' gradients: the focal term is a mean over the points (averages of halves = the whole), the offset
        # terms are SUMS over the frames (average of halves = half the whole)
        return 2.0 * terms["loss_rgbd_seg"] + offsets_scale * (terms["loss_kp_of"] + terms["loss_ctr_of"])

    names = ["rndla_ds_stages.0.lfa.att_pooling_1.fc.weight", "cnn_ds_stages.3.0.stages.1.1.weight", "kp_ofst_layer.3.conv.weight",
             "ds_fuse_p2r_pre_layers.0.conv.weight", "rndla_up_stages.2.conv.weight", "cnn_pre_stages.1.weight", "ctr_ofst_layer.0.conv.weight"]

    def grads_of(module, inputs, targets, scale):
        module.zero_grad()
        objective(module(inputs), targets, scale).backward()
        params = dict((n.replace("module.", "", 1) if n.startswith("module.") else n, p) for n, p in module.named_parameters())
        return {n: params[n].grad.detach().float().flatten()[:256].cpu().numpy().tolist() for n in names}

    def bn_eval(module):
        # Batch statistics of 8 samples per channel (4 points x 2 frames at the deepest level) amplify rounding noise to percents of a
        # gradient (two runs of ONE process differ by 1-30 %%): the gradient comparison uses the running statistics; what SyncBatchNorm
        # synchronises is checked on the first BatchNorm of each branch, whose batch statistics depend on the inputs only
        for m in module.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()

    FIRST_BN = ["cnn_pre_stages.1", "rndla_pre_stages.bn.bn"]

    def first_bn_stats(module, inputs):
        with torch.no_grad():
            module(inputs)                                     # train(): one update of the running statistics, momentum 0.1
        mods = dict((n.replace("module.", "", 1) if n.startswith("module.") else n, m) for n, m in module.named_modules())
        return {n: [mods[n].running_mean.float().cpu().numpy().tolist(), mods[n].running_var.float().cpu().numpy().tolist()] for n in FIRST_BN}

    net = build()
    inputs, targets = batch(rank * PER, PER)
    wrapped = D.wrap_ddp(net, dev) if world > 1 else net       # RCCL: SyncBatchNorm + DDP
    sync_bn = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in wrapped.modules())
    stats = first_bn_stats(wrapped, inputs)
    bn_eval(wrapped)
    with D.count_collectives() as cc:                          # Python-side collectives of one training step (SyncBatchNorm's)
        got = grads_of(wrapped, inputs, targets, 1.0)
    out = dict(rank=rank, world=world, backend=(dist.get_backend() if world > 1 else "none"), sync_bn=sync_bn, got=got, stats=stats,
               rccl_ranks=(dist.get_world_size() if world > 1 else 1), collectives=cc.counts,
               ddp_buckets=(D.ddp_bucket_count(wrapped) if world > 1 else None))
    if rank == 0:                                              # ONE process on all world * PER frames, plain BatchNorm
        ref = build()
        inputs_all, targets_all = batch(0, world * PER)
        out["want_stats"] = first_bn_stats(ref, inputs_all)
        bn_eval(ref)
        out["want"] = grads_of(ref, inputs_all, targets_all, 1.0 / world)
    print("RESULT " + json.dumps(out), flush=True)
    g.close()
""") % (ROOT, ROOT)


def _run_worker(tmp_path, world):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    procs, logs = [], []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = open(tmp_path / ("rank%d.log" % rank), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=log, stderr=subprocess.STDOUT, text=True))
    try:
        for p in procs:
            p.wait(timeout=900)
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
    res = [json.loads([ln for ln in o.splitlines() if ln.startswith("RESULT ")][0][7:]) for o in outs]
    return sorted(res, key=lambda r: r["rank"])


def _check(res):
    want = res[0]["want"]
    for name, w in want.items():
        w = np.array(w)
        assert np.abs(w).max() > 0, name
        for r in res:      # two runs of one process differ by 1e-4 .. 3e-3 of a gradient's norm (MIOpen's algorithms, float atomics)
            err = np.linalg.norm(np.array(r["got"][name]) - w) / np.linalg.norm(w)
            assert err <= 1e-2, (name, r["rank"], err)
    for name, (mean, var) in res[0]["want_stats"].items():     # batch statistics of the WHOLE batch reached every rank's running statistics
        for r in res:
            np.testing.assert_allclose(r["stats"][name][0], mean, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(mean).max()), err_msg=name)
            np.testing.assert_allclose(r["stats"][name][1], var, rtol=1e-3, atol=1e-6, err_msg=name)


def test_gradient_worker_one_rank_rehearsal(tmp_path):
    """the worker of the two-rank check with one rank: no collective, plain BatchNorm -- its own reference on the same two frames"""
    res = _run_worker(tmp_path, 1)
    assert res[0]["world"] == 1 and res[0]["sync_bn"] == 0 and res[0]["collectives"] == {} and res[0]["rccl_ranks"] == 1
    _check(res)


@two_gpus
def test_two_rccl_ranks_leave_the_gradients_of_the_doubled_batch(tmp_path):
    res = _run_worker(tmp_path, 2)
    assert [r["world"] for r in res] == [2, 2] and all(r["backend"] == "nccl" for r in res)
    assert all(r["sync_bn"] > 50 for r in res)                 # BatchNorm layers were converted (train_lm.py:592)
    # first contact with a multi-GPU node: what ran (read with pytest -s / in the failure report)
    print("RCCL ranks:", [r["rccl_ranks"] for r in res], "SyncBatchNorm layers:", res[0]["sync_bn"],
          "python-side collectives of one step (BatchNorm in eval for the gradient check: none expected here):", res[0]["collectives"],
          "DDP gradient buckets:", res[0]["ddp_buckets"])
    assert all(r["rccl_ranks"] == 2 for r in res)
    _check(res)
    for name in res[0]["got"]:                                 # one gradient on both ranks, to the last bits the all-reduce leaves
        np.testing.assert_allclose(res[0]["got"][name], res[1]["got"][name], rtol=1e-6, atol=0, err_msg=name)
