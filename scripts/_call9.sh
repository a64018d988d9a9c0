cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t9.txt 2>&1
tail -6 gpurun_out/t9.txt
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
timeout 500 python bench.py --steps 10 --warmup 3 --config 4 --no-cpu-baseline > gpurun_out/r02_bench_config4.json 2> gpurun_out/r02_bench_config4.err
timeout 600 python bench.py --steps 5 --warmup 2 --mode train --no-cpu-baseline > gpurun_out/r02_bench_train.json 2> gpurun_out/r02_bench_train.err
python -c "
import json
for f in ('r02_bench_default','r02_bench_config4','r02_bench_train'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],1), round(d['ms_per_step'],2), d['breakdown_ms'], d['roofline']['kernel'][:40], round(d['roofline']['frac'],3), d.get('cpu_baseline',{}).get('value'))
        print('   knn', d['hot_path_ops'].get('knn'))
    except Exception as e: print(f, 'FAILED', e)
"
tail -3 gpurun_out/r02_bench_train.err
