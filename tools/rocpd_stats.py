#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (default output of `rocprofv3 --kernel-trace`) as a
per-kernel table: calls, total / average / min / max duration (us), % of GPU kernel time.

    python tools/rocpd_stats.py gpurun_out/prof/r_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=60):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute(
        "select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col)
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        print("%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (
            r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total))
    print("TOTAL kernel time: %.1f us over %d dispatches" % (total / 1e3, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
