#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: every dispatch with its start offset, duration, queue / stream and the idle
time of ITS queue in front of it; per-queue busy time.  The step = the dispatches between two consecutive launches of --anchor
(a kernel that runs once per step), counted from the end of the marked region backwards.

    python scripts/rocpd_timeline.py /tmp/prof_k/k_results.db --between check_range_kernel --anchor affine_relu_maxpool --step 2
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--between", default="check_range_kernel")
    ap.add_argument("--anchor", default="affine_relu_maxpool")
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--width", type=int, default=70)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    print("# columns of the kernels view:", ", ".join(cols))
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    lo, hi = db.execute("select min(end), max(start) from kernels where name like ?", (f"%{a.between}%",)).fetchone()
    sel = f"name, start, end, {qcol}" if qcol else "name, start, end, 0"
    rows = db.execute(f"select {sel} from kernels where start >= {lo} and end <= {hi} order by start").fetchall()
    anchors = [i for i, r in enumerate(rows) if a.anchor in r[0]]
    print(f"# {len(rows)} dispatches in the marked region, {len(anchors)} anchors ({a.anchor}); stream column: {qcol}")
    # a step starts at the first dispatch after the previous step's last one: use the anchor of step s and s+1 and cut at the
    # largest all-queue idle point between them is overkill -- print from anchor s to anchor s+1 (one full period)
    i0, i1 = anchors[a.step], anchors[a.step + 1]
    t0 = rows[i0][1]
    last_end = {}
    busy = {}
    print("%10s %9s %8s %4s  %s" % ("start_us", "dur_us", "q_idle", "q", "kernel"))
    for name, s, e, q in rows[i0:i1]:
        idle = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        busy[q] = busy.get(q, 0) + (e - s)
        short = name.replace("void ", "").replace("ffb6d::(anonymous namespace)::", "").replace("ffb6d::", "")
        print("%10.1f %9.1f %8.1f %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle, q, short[: a.width]))
    span = (rows[i1][1] - t0) / 1e3
    print(f"# period {span:.1f} us; busy per queue:", {q: round(b / 1e3, 1) for q, b in busy.items()})


if __name__ == "__main__":
    main()
