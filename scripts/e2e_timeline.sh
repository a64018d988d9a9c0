#!/bin/bash
# Kernel timeline of the overlapped sensor -> pose pipeline (scripts/probes/e2e_probe.py: input assembly of batch i + 1 and pose solver of batch i
# on side streams under the forward of batch i + 1):  bash scripts/e2e_timeline.sh [tag]  -> gpurun_out/<tag>_e2e_timeline.txt
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_e2e && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_e2e -o k -- python "$REPO/scripts/probes/e2e_probe.py" --iters 5 > "$OUT/${TAG}_e2e_probe_under_rocprof.txt" 2> "$OUT/e2e_prof.err" )
DB=$(find /tmp/prof_e2e -name '*.db' | head -1)
{ echo "# rocprofv3 --kernel-trace -- python scripts/probes/e2e_probe.py --iters 5; one period of the overlapped schedule (from one depth_to_cloud launch to the next)"
  python scripts/rocpd_timeline.py "$DB" --between depth_normal --anchor depth_to_cloud --step 6 --width 60 | python -c "
import sys
rows = sys.stdin.read().splitlines()
# the full listing is ~800 lines: keep the header, per-queue summary and a condensed view (launches >= 150 us and the pose / input kernels)
for r in rows:
    f = r.split()
    keep = r.startswith('#') or r.startswith('  start') or 'start_us' in r
    if not keep and len(f) >= 5:
        try:
            keep = float(f[1]) >= 150.0 or any(k in r for k in ('mean_shift', 'vote_sets', 'depth_', 'sample_', 'best_fit', 'knn_'))
        except ValueError:
            pass
    if keep:
        print(r)
"; } > "$OUT/${TAG}_e2e_timeline.txt" 2>&1
tail -5 "$OUT/${TAG}_e2e_timeline.txt"
