cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_forward_gpu.py -x -q -s > gpurun_out/t4.txt 2>&1
tail -5 gpurun_out/t4.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --trace-all > gpurun_out/r02_pm_bench.json 2> gpurun_out/r02_pm_trace.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --layout cm > gpurun_out/r02_cm_bench.json 2> gpurun_out/r02_cm_trace.txt
python -c "
import json
for f in ('r02_pm_bench','r02_cm_bench'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['breakdown_ms'], d['roofline']['kernel'], d['roofline']['frac'])
    except Exception as e: print(f, 'FAILED', e)
"
