#!/bin/bash
# Round-end records on the GPU box, all from HEAD's defaults:  bash scripts/round_end.sh r06   (through gpurun; ~15 minutes)
#   1. scripts/profile_round.sh: kernel stats + PMC FETCH_SIZE / WRITE_SIZE passes of the default bench
#   2. the default bench line with cpu_baseline                                   -> <tag>_bench_default_run.json
#   3. configurations 4 and 5: bench line with cpu_baseline + rocprofv3 kernel stats -> <tag>_bench_config{4,5}_run.json, <tag>_kernels_config{4,5}.txt
#   4. per-op / per-shape table on one stream                                      -> <tag>_hot_path_ops_by_shape_one_stream.txt
#   5. fused LFA: level table and PMC of the level-0 launch; MFMA-busy PMC of the dominant GEMM
#   6. training step (bf16 autocast and fp32; MIOpen's cold start alone is ~100 s: generous limits) + its kernel statistics
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
bash scripts/profile_round.sh "$TAG" > /dev/null
bash scripts/profile_round.sh "${TAG}_config5" --config 5 > /dev/null            # PMC traffic of the bf16 workload (roofline.traffic of its line)
timeout 300 python bench.py > "$OUT/${TAG}_bench_default_run.json" 2> "$OUT/${TAG}_bench_default.err"
for C in 4 5; do
    timeout 300 python bench.py --config $C > "$OUT/${TAG}_bench_config${C}_run.json" 2> "$OUT/${TAG}_bench_config${C}.err"
    ( cd /tmp && rm -rf /tmp/prof_c$C && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c$C -o k -- python "$REPO/bench.py" --config $C --steps 5 --warmup 3 \
          --no-cpu-baseline --mark-region > /dev/null 2> "$OUT/${TAG}_prof_c$C.err"
      DB=$(find /tmp/prof_c$C -name '*.db' | head -1)
      { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config $C --steps 5 --warmup 3 --no-cpu-baseline --mark-region"
        echo "# timed region only (between the two check_range_kernel markers), per-step = totals / 5"
        python "$REPO/scripts/rocpd_stats.py" "$DB" --between check_range_kernel --steps 5 --top 40; } > "$OUT/${TAG}_kernels_config$C.txt" 2>&1 )
done
{ echo "# python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --trace-all   (one stream: clean per-kernel durations; bs=8, N=12288, fp32)"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --trace-all 2>&1 >/dev/null | grep -v amdgpu.ids; } > "$OUT/${TAG}_hot_path_ops_by_shape_one_stream.txt"
timeout 200 python scripts/bench_lfa.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_lfa_levels.txt"
bash scripts/pmc_lfa.sh 0 1 f32 > /dev/null 2>&1; cp "$OUT/lfa_pmc_0_1_f32.txt" "$OUT/${TAG}_lfa_pmc_level0_half1.txt"
bash scripts/pmc_pm_shape.sh 1024 2304 38400 f32 0 > /dev/null 2>&1; cp "$OUT/pm_shape_pmc_1024_2304_38400_f32_0.txt" "$OUT/${TAG}_mlp_pm_seq_pmc_1024_2304_38400.txt"
for P in bf16 fp32; do
    timeout 400 python bench.py --mode train --precision $P --steps 8 --warmup 3 --no-cpu-baseline --cudnn-benchmark 0 \
        > "$OUT/${TAG}_bench_train_$P.json" 2> "$OUT/${TAG}_bench_train_$P.err"
done
timeout 300 bash scripts/prof_train.sh --precision bf16 > /dev/null 2>&1; cp "$OUT/train_kernels.txt" "$OUT/${TAG}_rocprofv3_kernel_stats_train_bf16.txt"; cp "$OUT/train_under_rocprof.json" "$OUT/${TAG}_bench_train_bf16_under_rocprof.json"
# round 5: exact-KNN PMC, pose solver (bench + kernel statistics), input pipeline
bash scripts/pmc_knn.sh "$TAG" > /dev/null 2>&1
timeout 120 python scripts/pyramid_loop.py --iters 20 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_pyramid_alone.txt"
timeout 200 python scripts/bench_pose.py > "$OUT/${TAG}_pose_bench.json" 2> "$OUT/${TAG}_pose_bench.err"
( cd /tmp && rm -rf /tmp/prof_pose && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_pose -o k -- python "$REPO/scripts/bench_pose.py" --steps 4 > /dev/null 2> "$OUT/${TAG}_pose_prof.err"
  DB=$(find /tmp/prof_pose -name '*.db' | head -1); python "$REPO/scripts/rocpd_stats.py" "$DB" --top 25 > "$OUT/${TAG}_pose_kernels.txt" 2>&1 )
timeout 120 python scripts/bench_inputs.py > "$OUT/${TAG}_inputs_bench.json" 2> "$OUT/${TAG}_inputs_bench.err"
# round 6: sensor -> pose pipeline (serial / overlapped), operator-level drop-in, bf16 GEMM and upconv form tables, config-5 GEMM PMC
timeout 300 python bench.py --mode e2e --steps 20 > "$OUT/${TAG}_bench_e2e.json" 2> "$OUT/${TAG}_bench_e2e.err"
timeout 400 python bench.py --path dropin --steps 10 > "$OUT/${TAG}_bench_dropin.json" 2> "$OUT/${TAG}_bench_dropin.err"
timeout 200 python scripts/probes/big_gemm_probe.py --hints 7,9 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_big_gemm_forms.txt"
timeout 200 python scripts/probes/upconv_probe.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_upconv_forms.txt"
bash scripts/pmc_pm_shape.sh 1024 1024 76800 bf16 9 > /dev/null 2>&1; cp "$OUT/pm_shape_pmc_1024_1024_76800_bf16_9.txt" "$OUT/${TAG}_mlp_pm_big_pmc_1024_1024_76800.txt"
python -c "
import json
for n in ('default', 'config4', 'config5'):
    p = json.load(open('$OUT/${TAG}_bench_%s_run.json' % n))
    print(n, round(p['value'], 1), 'frames/s', round(p['ms_per_step'], 2), 'ms', p['roofline']['kernel'], round(p['roofline']['frac'], 3), 'traffic', p['roofline']['traffic'], 'cpu', p.get('cpu_baseline', {}).get('value'))
"
python -c "
import json
e = json.load(open('$OUT/${TAG}_bench_e2e.json'))['e2e']['synthetic_votes']
print('e2e serial', round(e['serial']['ms_per_batch'], 2), 'overlapped', round(e['overlapped']['ms_per_batch'], 2), e['stage_ms'])
d = json.load(open('$OUT/${TAG}_bench_dropin.json'))
print('dropin', round(d['ms_per_step'], 2), 'ms; plain torch operators', round(d['same_forward_with_plain_torch_operators']['ms_per_step'], 2))
"
