cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/round_start_ab.sh > /dev/null 2>&1
timeout 300 python scripts/hbm_probe.py > gpurun_out/hbm_probe.txt 2>&1
timeout 400 python scripts/cnn_layout_probe.py > gpurun_out/cnn_layout_probe.txt 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --trace-all > gpurun_out/r02_base_bench.json 2> gpurun_out/r02_base_trace.txt
tail -5 gpurun_out/ab_lds.txt
