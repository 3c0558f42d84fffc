#!/usr/bin/env python3
"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value), max(value) from counters_collection "
                  "group by kernel_name, counter_name order by sum(value) desc").fetchall()
print("%-60s %-12s %6s %16s %16s" % ("kernel", "counter", "calls", "sum", "max_per_call"))
for name, ctr, calls, tot, mx in rows:
    short = name.split("(")[0].replace("void psacx::", "")
    print("%-60s %-12s %6d %16.1f %16.1f" % (short[:60], ctr, calls, tot, mx))
