#!/usr/bin/env python3
"""Prints per-kernel totals from a rocprofv3 rocpd database (kernel-trace): calls, total and average duration in
microseconds, share of the GPU time.  Durations come from the raw dispatch records (start / end in ns)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
try:
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    rows = [(n, c, t / 1e3, a / 1e3, 100.0 * t / total) for n, c, t, a in rows]
except sqlite3.Error:
    raw = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    rows = [(n, c, t / 1e3, a / 1e3, p) for n, c, t, a, p in raw]
print("%-70s %6s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void psacx::", "")
    print("%-70s %6d %12.1f %12.2f %6.2f" % (short[:70], calls, tot, avg, pct))
