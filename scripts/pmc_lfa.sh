#!/bin/bash
# PMC characterisation of one fused LFA launch:  bash scripts/pmc_lfa.sh LEVEL MODE [f32|bf16] [p_hint]
#   -> gpurun_out/lfa_pmc_<args>.txt   (separate --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
ARGS="$*"; TAG=$(echo "$ARGS" | tr ' ' '_')
PASSES=("SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
        "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
        "FETCH_SIZE"
        "WRITE_SIZE")
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python scripts/bench_lfa_one.py $ARGS  (12 launches; FETCH/WRITE_SIZE unit 1024 B)"
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1)); D=/tmp/pl$i; rm -rf $D
    timeout 120 rocprofv3 --kernel-trace --pmc $P -d $D -o p -- python "$REPO/scripts/bench_lfa_one.py" $ARGS > "$OUT/pl$i.out" 2> "$OUT/pl$i.err"
    DB=$(find $D -name '*.db' 2>/dev/null | head -1)
    if [ -n "$DB" ]; then
      python "$REPO/scripts/rocpd_pmc.py" "$DB" --match lfa_pm | cut -c1-60,90-140
      [ $i = 1 ] && cat "$OUT/pl1.out"
    else echo "# pass $i ($P) produced no database"; tail -2 "$OUT/pl$i.err"; fi
  done; } > "$OUT/lfa_pmc_$TAG.txt" 2>&1
cat "$OUT/lfa_pmc_$TAG.txt"
