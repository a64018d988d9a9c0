#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter values per kernel from a rocpd SQLite database.

    python scripts/rocpd_pmc.py /tmp/pmc/x_results.db [--between check_range_kernel] [--match ffb6d]

Prints kernel, counter, dispatches, mean value per dispatch.  FETCH_SIZE / WRITE_SIZE are in
KiB-like units of 1024 B on this rocprofv3 (MI355X_MICROARCH.md section HBM: multiply FETCH_SIZE
by 2 for wide coalesced reads on gfx950 before comparing with byte counts).
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="ffb6d")
    ap.add_argument("--schema", action="store_true")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    src = "counters_collection" if "counters_collection" in views else "pmc_events"
    cols = [r[1] for r in db.execute(f"pragma table_info({src})")]
    if args.schema:
        print(src, cols)
        for r in db.execute(f"select * from {src} limit 3"):
            print(r)
        return
    kcol = next(c for c in cols if c in ("kernel_name", "name", "kernel"))
    ccol = next(c for c in cols if c in ("counter_name", "pmc_name", "counter"))
    vcol = next(c for c in cols if c in ("value", "counter_value"))
    dcol = next((c for c in cols if c in ("dispatch_id", "event_id", "id")), None)
    q = (f"select {kcol}, {ccol}, count(distinct {dcol}), sum({vcol}) from {src} "
         f"where {kcol} like ? group by {kcol}, {ccol} order by 4 desc")
    print("%-90s %-14s %10s %18s" % ("kernel", "counter", "dispatches", "mean per dispatch"))
    for k, c, n, v in db.execute(q, (f"%{args.match}%",)):
        print("%-90s %-14s %10d %18.1f" % (k[:90], c, n, v / max(n, 1)))


if __name__ == "__main__":
    main()
