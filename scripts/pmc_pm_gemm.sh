#!/bin/bash
# PMC characterisation of the largest point-major shared-MLP GEMM (1024->1024, 8 x 4800 px) on the GPU box:
#   bash scripts/pmc_pm_gemm.sh [tile_hint] -> gpurun_out/pm_gemm_pmc.txt   (two separate --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
H=${1:-0}
rm -rf /tmp/pg1 /tmp/pg2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES \
    -d /tmp/pg1 -o p -- python "$REPO/scripts/bench_mlp_pm_one.py" $H > /dev/null 2> "$OUT/pg1.err"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum \
    -d /tmp/pg2 -o p -- python "$REPO/scripts/bench_mlp_pm_one.py" $H > /dev/null 2> "$OUT/pg2.err"
rm -rf /tmp/pg3
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d /tmp/pg3 -o p -- python "$REPO/scripts/bench_mlp_pm_one.py" $H > /dev/null 2> "$OUT/pg3.err"
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python scripts/bench_mlp_pm_one.py $H  (10 launches of 1024->1024 on 8 x 4800 px)"
  for d in /tmp/pg1 /tmp/pg2 /tmp/pg3; do
    DB=$(find $d -name '*.db' | head -1)
    python "$REPO/scripts/rocpd_pmc.py" "$DB" --match mlp_pm
    python "$REPO/scripts/rocpd_stats.py" "$DB" --top 3 | grep -i "mlp_pm\|kernel " | cut -c1-60,112-190
  done; } > "$OUT/pm_gemm_pmc_h$H.txt" 2>&1
cat "$OUT/pm_gemm_pmc_h$H.txt" | cut -c1-170
tail -3 $OUT/pg2.err $OUT/pg3.err | cut -c1-200
