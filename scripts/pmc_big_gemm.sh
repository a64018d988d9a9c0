#!/bin/bash
# PMC characterisation of the largest shared-MLP GEMM (1024->1024, 8 x 4800 px) on the GPU box:
#   bash scripts/pmc_big_gemm.sh  -> gpurun_out/big_gemm_pmc.txt   (two separate --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp || exit 1
rm -rf /tmp/pg1 /tmp/pg2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    -d /tmp/pg1 -o p -- python "$REPO/scripts/bench_mlp_one.py" > /dev/null 2> "$OUT/pg1.err"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES \
    -d /tmp/pg2 -o p -- python "$REPO/scripts/bench_mlp_one.py" > /dev/null 2> "$OUT/pg2.err"
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python scripts/bench_mlp_one.py   (10 launches of 1024->1024 on 8 x 4800 px)"
  for d in /tmp/pg1 /tmp/pg2; do
    DB=$(find $d -name '*.db' | head -1)
    python "$REPO/scripts/rocpd_pmc.py" "$DB" --match shared_mlp
    python "$REPO/scripts/rocpd_stats.py" "$DB" --top 3 | grep -i "shared_mlp\|kernel " | cut -c1-60,112-190
  done; } > "$OUT/big_gemm_pmc.txt" 2>&1
cat "$OUT/big_gemm_pmc.txt" | cut -c1-170
