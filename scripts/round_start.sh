#!/bin/bash
# First GPU call of a round:  bash scripts/round_start.sh r04        (about 6 GPU-minutes)
#   1. what round 3 left without a wall-clock number: the bf16 training step after the atomic-free gather backward
#      (csrc/train_rows.hip gather_sum_rows).  MIOpen's cold-cache start-up of the training convolutions alone is ~100 s on a
#      fresh box: the limit is 400 s on purpose (round 3 lost its last two bench calls to `timeout 100` / `timeout 120`).
#   2. the same step in fp32
#   3. kernel statistics of the bf16 step (scripts/prof_train.sh) and of the default inference bench (scripts/profile_round.sh)
TAG=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
timeout 400 python bench.py --mode train --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --cudnn-benchmark 0 \
    > "$OUT/${TAG}_start_bench_train_bf16.json" 2> "$OUT/${TAG}_start_bench_train_bf16.err"
cut -c1-260 "$OUT/${TAG}_start_bench_train_bf16.json"
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --cudnn-benchmark 0 \
    > "$OUT/${TAG}_start_bench_train_fp32.json" 2> "$OUT/${TAG}_start_bench_train_fp32.err"
cut -c1-260 "$OUT/${TAG}_start_bench_train_fp32.json"
timeout 300 bash scripts/prof_train.sh --precision bf16 | head -40
bash scripts/profile_round.sh "${TAG}_start"
timeout 300 python scripts/train_copy_census.py > "$OUT/${TAG}_train_copy_census.txt" 2>&1; head -30 "$OUT/${TAG}_train_copy_census.txt"
# the reference's own default geometry (common.py: 12800 points -> 12800 / 3200 / 800 / 200 / 50, ragged against every tile size): parity on the
# device before it becomes a case of tests/test_forward_gpu.py (round 3 checked the ragged paths on the emulator only)
timeout 200 python -c "
import sys; sys.path.insert(0, 'tests')
import torch, test_forward_gpu as T
dev = torch.device('cuda:0')
T.test_hot_path_matches_plain_torch_on_the_same_device(dev, (2, 1, 12800, 480, 640, 22)); print('N=12800 forward parity ok')
T.test_training_step_gradients_match_plain_torch(dev, n_pts=1100, height=136, width=168); print('ragged training gradients ok')
" 2>&1 | tail -5
FFB6D_LOGSOFTMAX_ROWS=1 FFB6D_GATHER_SUM_LANES=1 timeout 400 python bench.py --mode train --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --cudnn-benchmark 0 > "$OUT/${TAG}_start_bench_train_bf16_optin.json" 2> /dev/null; cut -c1-200 "$OUT/${TAG}_start_bench_train_bf16_optin.json"   # A/B of the row LogSoftmax + multi-lane gather backward
FFB6D_LOGSOFTMAX_ROWS=1 FFB6D_GATHER_SUM_LANES=1 FFB6D_BN_ROWS=1 timeout 400 python bench.py --mode train --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --cudnn-benchmark 0 > "$OUT/${TAG}_start_bench_train_bf16_bn_rows.json" 2> /dev/null; cut -c1-200 "$OUT/${TAG}_start_bench_train_bf16_bn_rows.json"   # A/B of the opt-in BatchNorm on rows
