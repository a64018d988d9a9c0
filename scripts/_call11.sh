cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pm_gpu.py tests/test_inputs_gpu.py -x -q > gpurun_out/t11.txt 2>&1
tail -12 gpurun_out/t11.txt
timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -s -k "bf16 or fusion_stage or weight_updates" > gpurun_out/t11b.txt 2>&1
grep "bf16 max err" gpurun_out/t11b.txt | head -40; tail -5 gpurun_out/t11b.txt
timeout 300 python bench.py --steps 10 --warmup 3 --config 5 --no-cpu-baseline > gpurun_out/r02_bench_config5.json 2> gpurun_out/r02_bench_config5.err
python -c "
import json
for f in ('r02_bench_config5',):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],1), round(d['ms_per_step'],2), d['breakdown_ms'], d['roofline']['kernel'][:50], round(d['roofline']['frac'],3))
        for k,v in sorted(d['hot_path_ops'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]: print('   ',k, round(v['ms_per_step'],3), round(v['algorithmic_GBps']), round(v.get('algorithmic_TFLOPs',0),1))
    except Exception as e: print(f, 'FAILED', e)
"
tail -3 gpurun_out/r02_bench_config5.err
