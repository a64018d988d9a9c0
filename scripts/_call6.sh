cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_pm_gpu.py -x -q -k "glue" 2>&1 | tail -2
for ov in 0 1; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --overlap-pyramid $ov > gpurun_out/r02b_ov$ov.json 2> gpurun_out/r02b_ov$ov.err
done
python -c "
import json
for f in ('r02b_ov0','r02b_ov1'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['breakdown_ms'], d['hot_path_ops'].get('psp_pool_pm'))
    except Exception as e: print(f, 'FAILED', e)
"
