#!/bin/bash
# Does MIOpen's split-K choice for the 3x3 convolutions pay inside the step?  Its igemm kernels split K over 2-8 workgroups for the
# 128..512-channel layers (tuned alone on the chip), which costs a zero-fill launch per convolution (SubTensorOpWithScalar1d, 0.36 ms per
# step) and float atomics (results not bit-reproducible).  Runs the bench with the pinned perf-db and with the same perf-db with
# gemm_k_global_split = 0 everywhere.   bash scripts/miopen_split_probe.sh   -> gpurun_out/r05_miopen_split_probe.txt
cd "$(dirname "$0")/../.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
run() {
    local tag=$1 db=$2
    ( cd /tmp && rm -rf /tmp/prof_$tag /tmp/mdb_$tag && mkdir -p /tmp/mdb_$tag && cp $db/*.txt /tmp/mdb_$tag/ && export MIOPEN_USER_DB_PATH=/tmp/mdb_$tag
      timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o k -- python "$REPO/bench.py" --steps 10 --warmup 5 --no-cpu-baseline --mark-region > "$OUT/split_$tag.json" 2> "$OUT/split_$tag.err"
      DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
      echo "== $tag"
      python -c "
import json
p = json.load(open('$OUT/split_$tag.json'))
print('   under rocprofv3:', round(p['value'], 1), 'frames/s', round(p['ms_per_step'], 3), 'ms')"
      python "$REPO/scripts/rocpd_stats.py" "$DB" --between check_range_kernel --steps 10 --top 40 | grep -i "igemm\|SubTensor\|per step" | cut -c1-100,112-170
      timeout 200 python "$REPO/bench.py" --steps 20 --no-cpu-baseline > "$OUT/split_${tag}_plain.json" 2>/dev/null
      python -c "
import json
p = json.load(open('$OUT/split_${tag}_plain.json'))
print('   plain run:', round(p['value'], 1), 'frames/s', round(p['ms_per_step'], 3), 'ms')" )
}
{ run pinned "$REPO/ffb6d_amd/miopen_pin"; run nosplit "$REPO/scripts/probes/miopen_nosplit"; } > "$OUT/r05_miopen_split_probe.txt" 2>&1
cat "$OUT/r05_miopen_split_probe.txt"
