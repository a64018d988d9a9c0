#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun):  bash scripts/profile_round.sh r01
#   1. rocprofv3 --kernel-trace --stats over `python bench.py --steps 5 --warmup 3 --no-cpu-baseline
#      --mark-region` (marker kernels cut the warm-up out)     -> gpurun_out/<tag>_kernels_steady.txt
#   2. rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes, kernel-trace only) over a
#      short bench run                                           -> gpurun_out/<tag>_pmc_{FETCH,WRITE}_SIZE.txt
# The databases stay in /tmp (gpurun_out is capped at 64 MiB); copy the text files into profiles/.
TAG=${1:-r01}
shift; EXTRA="$*"          # further arguments go to bench.py (e.g. --config 5)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/prof_k /tmp/prof_f /tmp/prof_w
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python "$REPO/bench.py" --steps 5 --warmup 3 \
    --no-cpu-baseline --mark-region $EXTRA > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_prof_k.err"
DB=$(find /tmp/prof_k -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --mark-region $EXTRA"
  echo "# timed region only (between the two check_range_kernel markers), per-step = totals / 5"
  python "$REPO/scripts/rocpd_stats.py" "$DB" --between check_range_kernel --steps 5 --top 60; } > "$OUT/${TAG}_kernels_steady.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/prof_$( [ $C = FETCH_SIZE ] && echo f || echo w )
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d $D -o p -- python "$REPO/bench.py" --steps 2 --warmup 2 \
        --no-cpu-baseline --cudnn-benchmark 0 $EXTRA > /dev/null 2> "$OUT/${TAG}_prof_$C.err"
    DB=$(find $D -name '*.db' | head -1)
    { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --cudnn-benchmark 0 $EXTRA"
      echo "# unit: 1024 B; whole run (warm-up + timed steps), mean per dispatch"
      python "$REPO/scripts/rocpd_pmc.py" "$DB" --match ffb6d; } > "$OUT/${TAG}_pmc_$C.txt" 2>&1
done
tail -2 "$OUT/${TAG}_prof_k.err"
head -12 "$OUT/${TAG}_kernels_steady.txt" | cut -c1-180
head -6 "$OUT/${TAG}_pmc_FETCH_SIZE.txt" | cut -c1-160
