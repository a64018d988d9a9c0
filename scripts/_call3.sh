cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pm_gpu.py -x -q > gpurun_out/t3.txt 2>&1
tail -40 gpurun_out/t3.txt
