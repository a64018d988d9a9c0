cd /root/repo
python -m pytest tests/test_pose_gpu.py -x -q -s 2>&1 | grep -v Warning | tail -25
timeout 300 python scripts/bench_pose.py --steps 5 > gpurun_out/pose_bench.json 2> gpurun_out/pose_bench.err; tail -3 gpurun_out/pose_bench.err; cat gpurun_out/pose_bench.json
