#!/bin/bash
# PMC characterisation of one point-major GEMM shape:  bash scripts/pmc_pm_shape.sh K Cout rows [f32|bf16] [hint]
#   -> gpurun_out/pm_shape_pmc_<K>_<Cout>_<rows>_<dtype>.txt   (separate --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
ARGS="$*"; TAG=$(echo "$ARGS" | tr ' ' '_')
PASSES=("SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
        "FETCH_SIZE"
        "WRITE_SIZE"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU")
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python scripts/bench_mlp_pm_shape.py $ARGS  (10 launches; FETCH/WRITE_SIZE unit 1024 B)"
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1)); D=/tmp/ps$i; rm -rf $D
    timeout 120 rocprofv3 --kernel-trace --pmc $P -d $D -o p -- python "$REPO/scripts/bench_mlp_pm_shape.py" $ARGS > /dev/null 2> "$OUT/ps$i.err"
    DB=$(find $D -name '*.db' 2>/dev/null | head -1)
    if [ -n "$DB" ]; then
      python "$REPO/scripts/rocpd_pmc.py" "$DB" --match mlp_pm
      [ $i = 1 ] && python "$REPO/scripts/rocpd_stats.py" "$DB" --top 3 | grep -i "mlp_pm\|kernel " | cut -c1-60,112-190
    else echo "# pass $i ($P) produced no database"; tail -2 "$OUT/ps$i.err"; fi
  done; } > "$OUT/pm_shape_pmc_$TAG.txt" 2>&1
cut -c1-60,90-170 "$OUT/pm_shape_pmc_$TAG.txt"
