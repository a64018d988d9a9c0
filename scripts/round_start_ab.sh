#!/bin/bash
# First GPU call of the next round (one call, ~2 min): decide whether the experimental LDS-direct GEMM
# (csrc/shared_mlp.hip, FFB6D_MLP_PIPE=2) becomes the default.
#   1. parity of variant 2 on the shared-MLP and forward tests (same tests, switch in the environment)
#   2. A/B timing of variants 1 and 2 on the per-frame layer shapes (sum column must match between variants)
# Output: gpurun_out/ab_lds.txt
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== parity with FFB6D_MLP_PIPE=2"
  FFB6D_MLP_PIPE=2 timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x -k "shared_mlp or hot_path" 2>&1 | tail -3
  echo "== timing"
  FFB6D_MLP_AB_N=9 timeout 300 python scripts/bench_mlp_ab.py 1 2 2>&1 | grep -v "Warn\|amdgpu.ids"
} | tee gpurun_out/ab_lds.txt
