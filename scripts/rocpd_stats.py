#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace
--stats` on ROCm 7.2) as a per-kernel table: calls, total/avg/min/max duration, share of GPU time.

    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] > profiles/rNN_kernels.txt
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--width", type=int, default=110)
    ap.add_argument("--between", default=None,
                    help="only dispatches between the first and the last dispatch of a kernel whose name "
                         "contains this string (bench.py --mark-region launches check_range_kernel)")
    ap.add_argument("--steps", type=int, default=0, help="divide totals by this many steps")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    where = ""
    if args.between:
        lo, hi = db.execute(f"select min(end), max(start) from kernels where {name_col} like ?",
                            (f"%{args.between}%",)).fetchone()
        if lo is None or lo >= hi:
            raise SystemExit(f"marker kernel {args.between!r} not found twice")
        where = f"where start >= {lo} and end <= {hi}"
    rows = db.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels {where} group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {args.db}: {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels, "
          f"total kernel time {total / 1e6:.3f} ms")
    if args.steps:
        print(f"# per step ({args.steps} steps): {total / 1e6 / args.steps:.3f} ms of kernel time")
    print("%-*s %8s %12s %10s %10s %10s %7s" % (args.width, "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows[: args.top]:
        print("%-*s %8d %12.3f %10.1f %10.1f %10.1f %6.2f%%" % (
            args.width, name[: args.width], n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == "__main__":
    main()
