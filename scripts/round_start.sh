#!/bin/bash
# First GPU call of a round:  bash scripts/round_start.sh r03
#   1. kernel stats + PMC FETCH/WRITE passes of the default bench (scripts/profile_round.sh)
#   2. bench lines of configurations 4 and 5 (recorded before the folded up-convolution / fused position encoding)
#   3. A/B of what round 2 could not measure: tap-blend forms and the XCD-band workgroup order
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
bash scripts/profile_round.sh "$TAG"
for C in 4 5; do
    timeout 300 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench_config${C}_run.json" 2> "$OUT/${TAG}_bench_config${C}.err"
done
{ for F in static row select simple; do FFB6D_UPCONV_COMBINE=$F timeout 60 python scripts/blend_forms_ab.py; done
  echo "--- row-major workgroup order (FFB6D_UPCONV_XCD=0)"
  FFB6D_UPCONV_XCD=0 timeout 60 python scripts/blend_forms_ab.py; } > "$OUT/${TAG}_upconv_blend_forms_ab.txt" 2>&1
# what the last (GPU-less) hours of round 2 added, each against its switch
FFB6D_STEM_FUSED=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench_stem_unfused.json" 2> /dev/null
FFB6D_UPCONV_XCD=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench_blend_row_major_order.json" 2> /dev/null
timeout 300 python scripts/lds_probe.py > "$OUT/${TAG}_mlp_pm_lds_ab.txt" 2>&1      # bf16 LDS-tiled GEMM after the branch-free prefetch
FFB6D_UPCONV_FOLD=1 timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench_config5_folded.json" 2> /dev/null
tail -12 "$OUT/${TAG}_upconv_blend_forms_ab.txt"
