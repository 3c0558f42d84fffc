#!/usr/bin/env python3
"""Text summary of a rocprofv3 run stored as a rocpd SQLite database (this image's rocprofv3 writes <name>_results.db):
   tools/rocpd_summary.py <results.db> [kernel-name filter]
Kernel trace: calls, average / minimum / total duration per kernel.  PMC runs: every counter summed over its instances, averaged
over the dispatches of a kernel (SQ counters are per shader engine / XCD instance: the sum is the whole chip)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
                  f"where s.kernel_name like ? group by s.kernel_name order by 5 desc", ("%" + flt + "%",)).fetchall()
total = sum(r[4] for r in rows) or 1
print("%-8s %10s %10s %11s %6s  %s" % ("calls", "avg us", "min us", "total ms", "%", "kernel"))
for name, calls, avg, mn, tot in rows:
    print("%-8d %10.1f %10.1f %11.3f %6.1f  %s" % (calls, avg / 1e3, mn / 1e3, tot / 1e6, 100.0 * tot / total, name))
pm = db.execute(f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pe} e join {pi} i on e.pmc_id=i.id join {kd} d on e.event_id=d.event_id "
                f"join {ks} s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name, i.name order by 1, 2", ("%" + flt + "%",)).fetchall()
if pm:
    print("\ncounters: sum over instances, per dispatch (average over the dispatches)")
    for name, cn, val, nd in pm:
        print("%-28s %18.0f  (%d dispatches)  %s" % (cn, val / max(nd, 1), nd, name))
