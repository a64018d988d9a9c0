#!/bin/bash
# First GPU call of a round:  bash scripts/round_start.sh r03
#   1. kernel stats + PMC FETCH/WRITE passes of the default bench (scripts/profile_round.sh)
#   2. bench lines of configurations 4 and 5 at the round's starting HEAD (the round-end records are made by scripts/round_end.sh)
#   3. A/B of what round 2 left unmeasured: tap-blend forms / XCD-band order, stem fusion, bf16 LDS-tiled GEMM prefetch
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
bash scripts/profile_round.sh "${TAG}_start"
for C in 4 5; do
    timeout 300 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_start_bench_config${C}_run.json" 2> "$OUT/${TAG}_start_bench_config${C}.err"
done
{ for F in static select; do FFB6D_UPCONV_COMBINE=$F timeout 60 python scripts/blend_forms_ab.py; done
  echo "--- row-major workgroup order (FFB6D_UPCONV_XCD=0)"
  FFB6D_UPCONV_XCD=0 timeout 60 python scripts/blend_forms_ab.py; } > "$OUT/${TAG}_upconv_blend_forms_ab.txt" 2>&1
FFB6D_STEM_FUSED=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_start_bench_stem_unfused.json" 2> /dev/null
timeout 300 python scripts/lds_probe.py > "$OUT/${TAG}_start_mlp_pm_lds_ab.txt" 2>&1      # bf16 LDS-tiled GEMM after the branch-free prefetch
tail -12 "$OUT/${TAG}_upconv_blend_forms_ab.txt"
