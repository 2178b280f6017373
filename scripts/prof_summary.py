"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result as text: python scripts/prof_summary.py db [steps]"""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (durations in us; %d kernel names; total %.1f us; /step = total / %g steps)" % (len(rows), tot, steps))
print("%-10s %-12s %-10s %-8s %-10s %s" % ("calls", "total_us", "avg_us", "pct", "us/step", "kernel"))
for name, calls, total, avg, pct in rows:
    short = name.replace("void saunet::", "").replace("saunet::", "")
    if len(short) > 150:
        short = short[:150] + "..."
    print("%-10d %-12.1f %-10.2f %-8.2f %-10.1f %s" % (calls, total, avg, pct, total / steps, short))
