"""bench.py's own launch paths on the GPU box (the driver's contract): the multi-rank self-spawn (`--gpus N` re-executes the
script under torch.distributed.run, one process per rank) exercised with two ranks sharing the one GPU of the box over gloo
-- inference (batch sharding, no data-path collective, MAX over ranks) and training (DDP gradient all-reduce,
train_lm.py:559-563,625-628)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):           # a clean single-process start: bench.py spawns the ranks itself
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line:\n" + out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_spawns_two_ranks_and_reports_the_whole_job():
    line = _bench("--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--cudnn-benchmark", "0")
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert line["steps"] == 2 and line["warmup"] == 1 and line["unit"] == "frames/s"
    # whole-job aggregate: 16 frames per step over the slowest rank's time
    assert abs(line["value"] - 16 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert "roofline" in line and "gloo" in line["config"]["parallelism"]


def test_bench_train_mode_wraps_ddp_on_two_ranks():
    # one frame per rank and MIOpen's default algorithms: the launch path is what is under test, not the step time
    line = _bench("--gpus", "2", "--dist-backend", "gloo", "--mode", "train", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  "--batch", "1", "--cudnn-benchmark", "0")
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2
    assert "train" in line["metric"] and line["value"] > 0
