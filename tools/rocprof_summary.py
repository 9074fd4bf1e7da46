#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result into a small text table for profiles/.

usage: tools/rocprof_summary.py <results.db> [--pmc] > profiles/<name>.txt
Kernel-trace mode prints per-kernel calls / total / average / share (the `--stats` view);
--pmc prints per-kernel averages of every collected counter.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pmc = "--pmc" in sys.argv
    c = sqlite3.connect(db)
    if not pmc:
        rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
        print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
        for name, calls, total, avg, pct in rows:
            print(f"{name[:70]:<70} {calls:>6} {total/1e3:>10.3f} {avg:>10.2f} {pct:>6.2f}")
        return
    cols = [d[1] for d in c.execute('pragma table_info("counters_collection")')]
    sys.stderr.write("counters_collection columns: %s\n" % cols)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, avg(value), count(*) from counters_collection group by {name_col}, counter_name order by {name_col}").fetchall()
    cur = None
    for k, cn, v, n in rows:
        if k != cur:
            print(f"\n{k[:100]}  (dispatches: {n})")
            cur = k
        print(f"    {cn:<32} {v:>18.1f}")


if __name__ == "__main__":
    main()
