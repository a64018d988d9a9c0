cd $GRAFT_REPO_ROOT
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_k
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $REPO/bench.py --steps 5 --warmup 3 --no-cpu-baseline --mark-region > $OUT/r02a_bench_under_rocprof.json 2> $OUT/r02a_prof_k.err
DB=$(find /tmp/prof_k -name '*.db' | head -1)
python $REPO/scripts/rocpd_stats.py "$DB" --between check_range_kernel --steps 5 --top 70 > $OUT/r02a_kernels_steady.txt 2>&1
cd $REPO
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 --trace-all > $OUT/r02a_pm_1s_bench.json 2> $OUT/r02a_pm_1s_trace.txt
head -30 $OUT/r02a_kernels_steady.txt | cut -c1-160
