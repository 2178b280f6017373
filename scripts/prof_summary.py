"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result as text: python scripts/prof_summary.py db [steps]"""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (durations in us; %d kernel names; total %.1f us; /step = total / %g steps)" % (len(rows), tot, steps))
if "--families" in sys.argv:
    # all template instances of one kernel = one family; bench.py's roofline takes its headline kernel from the FIRST row of this table
    import collections
    fam = collections.OrderedDict()
    for name, calls, total, avg, pct in rows:
        short = name.replace("void saunet::", "").replace("saunet::", "")
        if "rocclr" in short or "at::native" in short:
            continue
        k = short.split("<")[0].split("(")[0]
        e = fam.setdefault(k, [0, 0.0, 0.0, 0])
        e[0] += calls; e[1] += total; e[2] += pct; e[3] += 1
    print("# [families]  (library kernels only; `kernel` = family name, instances = template instantiations in this trace)")
    print("%-10s %-12s %-10s %-8s %-10s %s" % ("calls", "total_us", "avg_us", "pct", "us/step", "kernel"))
    for k, (calls, total, pct, inst) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-10d %-12.1f %-10.2f %-8.2f %-10.1f %s   [%d instance%s]" % (calls, total, total / calls, pct, total / steps, k, inst, "" if inst == 1 else "s"))
    print("# [symbols]")
print("%-10s %-12s %-10s %-8s %-10s %s" % ("calls", "total_us", "avg_us", "pct", "us/step", "kernel"))
for name, calls, total, avg, pct in rows:
    short = name.replace("void saunet::", "").replace("saunet::", "")
    if len(short) > 150:
        short = short[:150] + "..."
    print("%-10d %-12.1f %-10.2f %-8.2f %-10.1f %s" % (calls, total, avg, pct, total / steps, short))
