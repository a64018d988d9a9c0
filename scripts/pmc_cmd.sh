#!/bin/bash
# PMC characterisation of the kernels one command launches:  bash scripts/pmc_cmd.sh <tag> <kernel-name substring> <command ...>
#   -> gpurun_out/pmc_<tag>.txt   (separate --pmc passes, kernel-trace only; FETCH_SIZE / WRITE_SIZE in their own passes)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
TAG=$1; MATCH=$2; shift 2
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
        "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
        "FETCH_SIZE"
        "WRITE_SIZE"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- $*   (FETCH/WRITE_SIZE unit 1024 B; gfx950: double FETCH_SIZE for wide coalesced reads)"
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1)); D=/tmp/pc$i; rm -rf $D
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $P -d $D -o p -- "$@" > /dev/null 2> "$OUT/pc$i.err" )
    DB=$(find $D -name '*.db' 2>/dev/null | head -1)
    if [ -n "$DB" ]; then
      python "$REPO/scripts/rocpd_pmc.py" "$DB" --match "$MATCH"
      [ $i = 1 ] && python "$REPO/scripts/rocpd_stats.py" "$DB" --top 6 | cut -c1-70,112-190
    else echo "# pass $i ($P) produced no database"; tail -2 "$OUT/pc$i.err"; fi
  done; } > "$OUT/pmc_$TAG.txt" 2>&1
