cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_pm_gpu.py -x -q 2>&1 | tail -2
MLP_PM_BIG=1 timeout 300 python scripts/bench_mlp_pm.py 1 2 8 9 2>&1 | grep -v amdgpu | cut -c1-330 > gpurun_out/mlp_pm_big.txt
cat gpurun_out/mlp_pm_big.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02c.json 2> gpurun_out/r02c.err
python -c "
import json
d=json.load(open('gpurun_out/r02c.json')); print(d['value'], d['ms_per_step'], d['breakdown_ms']); o=d['hot_path_ops']
for k in ('psp_pool_pm','bilinear_resize_pm'): print(k, o.get(k))
"
