cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pm_gpu.py tests/test_knn_gpu.py -x -q -k "pm or pick" > gpurun_out/t2.txt 2>&1
tail -15 gpurun_out/t2.txt
timeout 600 python scripts/bench_mlp_pm.py 0 1 2 4 5 > gpurun_out/mlp_pm_bench.txt 2>&1
