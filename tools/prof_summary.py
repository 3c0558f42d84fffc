#!/usr/bin/env python3
"""Prints per-kernel totals from a rocprofv3 rocpd database (kernel-trace)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("%-70s %6s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void psacx::", "")
    print("%-70s %6d %12.1f %12.2f %6.2f" % (short[:70], calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
