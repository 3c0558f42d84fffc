#!/usr/bin/env python3
"""Dispatches of a rocprofv3 kernel trace (rocpd SQLite database) in launch order: tools/rocpd_timeline.py <results.db> [min_us] [from_ms] [to_ms]
start (ms from the first dispatch), duration (us), gap to the dispatch before (us), kernel (template arguments cut)."""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t_from = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_to = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0 = rows[0][0] if rows else 0
prev_end = t0
for st, en, name in rows:
    ms = (st - t0) / 1e6
    gap = (st - prev_end) / 1e3
    prev_end = max(prev_end, en)
    if (en - st) / 1e3 < min_us or ms < t_from or ms > t_to:
        continue
    m = re.search(r"psacx(\d+)", name)
    short = name
    if m:
        L = int(m.group(1))
        short = name[m.end():m.end() + L]
        rest = name[m.end() + L:]
        if rest.startswith("I"):
            short += "<" + rest[1:41] + ">"
    print("%10.3f ms %10.1f us  gap %8.1f  %s" % (ms, (en - st) / 1e3, gap, short))
