#!/bin/bash
# PMC characterisation of the exact-KNN kernels inside one index pyramid:  bash scripts/pmc_knn.sh [tag]
#   -> gpurun_out/<tag>_knn_pmc.txt   (separate --pmc passes, kernel-trace only; 7 pyramids of 8 frames per pass)
TAG=${1:-r05}
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
        "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python scripts/pyramid_loop.py   (7 pyramids, bs = 8, N = 12288; SQ cycle counters in quad-cycles)"
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1)); D=/tmp/pk$i; rm -rf $D
    timeout 150 rocprofv3 --kernel-trace --pmc $P -d $D -o p -- python "$REPO/scripts/pyramid_loop.py" > "$OUT/pk$i.out" 2> "$OUT/pk$i.err"
    DB=$(find $D -name '*.db' 2>/dev/null | head -1)
    if [ -n "$DB" ]; then
      python "$REPO/scripts/rocpd_pmc.py" "$DB" --match knn | cut -c1-64,90-140
      [ $i = 1 ] && { cat "$OUT/pk1.out"; python "$REPO/scripts/rocpd_stats.py" "$DB" --top 14 | cut -c1-64,112-190; }
    else echo "# pass $i ($P) produced no database"; tail -2 "$OUT/pk$i.err"; fi
  done; } > "$OUT/${TAG}_knn_pmc.txt" 2>&1
cat "$OUT/${TAG}_knn_pmc.txt" | head -70
