#!/bin/bash
# kernel statistics of the training step:  bash scripts/prof_train.sh [extra bench flags]  -> gpurun_out/train_kernels.txt
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
rm -rf /tmp/prof_t
timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python "$REPO/bench.py" --mode train --steps 3 --warmup 2 --no-cpu-baseline \
    --cudnn-benchmark 0 --mark-region "$@" > "$OUT/train_under_rocprof.json" 2> "$OUT/train_prof.err"
DB=$(find /tmp/prof_t -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --cudnn-benchmark 0 --mark-region $*"
  python "$REPO/scripts/rocpd_stats.py" "$DB" --between check_range_kernel --steps 3 --top 45; } > "$OUT/train_kernels.txt" 2>&1
cut -c1-110,112-170 "$OUT/train_kernels.txt" | head -50
