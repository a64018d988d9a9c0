#!/bin/bash
# (round 4, profiles/r05_design_md_history_notes.md section 7): how much of the bench line is MIOpen's solver draw, and which deterministic setting is best.
#   bash scripts/miopen_draw_probe.sh   (through gpurun; ~5 minutes: immediate mode compiles its kernels on a cold cache)
# -> gpurun_out/miopen_draw_probe.txt: frames/s, ms per step and the convolution kernels of the 512-channel layers for
#    (a) find mode twice (the bench default; two processes = two draws), (b) immediate mode (--cudnn-benchmark 0),
#    (c) find mode with CK's grouped-convolution solver excluded.
cd "$(dirname "$0")/../.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
run() {   # tag, extra env (NAME=VALUE or -), bench flags...
    local tag=$1 envs=$2; shift 2
    ( cd /tmp && rm -rf /tmp/prof_$tag /tmp/mdb_$tag && mkdir -p /tmp/mdb_$tag && export MIOPEN_USER_DB_PATH=/tmp/mdb_$tag
      [ "$envs" != "-" ] && export "$envs"
      timeout ${TO:-240} rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o k -- python "$REPO/bench.py" --steps 10 --warmup 5 --no-cpu-baseline --mark-region "$@" \
          > "$OUT/miopen_draw_$tag.json" 2> "$OUT/miopen_draw_$tag.err"
      DB=$(find /tmp/prof_$tag -name '*.db' | head -1)
      echo "== $tag ($envs $*)"
      python -c "
import json
p = json.load(open('$OUT/miopen_draw_$tag.json'))
print('   under rocprofv3:', round(p['value'], 1), 'frames/s', round(p['ms_per_step'], 3), 'ms')"
      python "$REPO/scripts/rocpd_stats.py" "$DB" --between check_range_kernel --steps 10 --top 12 | cut -c1-70,112-170 | sed -n 4,17p )
}
# every run gets its own (empty) user find-db: otherwise the second process of a box re-reads the first one's draw
{ run find1 -; run find2 -; run find3 -; run nock MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS=0; run immediate - --cudnn-benchmark 0; } > "$OUT/miopen_draw_probe.txt" 2>&1
for t in find1 find2 find3 nock immediate; do mkdir -p "$OUT/miopen_db_$t"; cp /tmp/mdb_$t/*.txt "$OUT/miopen_db_$t/" 2>/dev/null; ls -la /tmp/mdb_$t >> "$OUT/miopen_draw_probe.txt"; done
cat "$OUT/miopen_draw_probe.txt"
