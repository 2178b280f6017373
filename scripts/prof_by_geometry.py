"""Per-(kernel, launch geometry) table of a rocprofv3 kernel trace (rocpd sqlite): python scripts/prof_by_geometry.py db steps [min_us_per_step]
Groups dispatches by kernel name + grid + workgroup + LDS size, i.e. by layer geometry -- the launch-mix view of the step."""
import collections
import re
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2])
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
con = sqlite3.connect(db)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, gx, gy, gz, wx, lds, vg, dur in con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, duration from kernels"):
    k = (name, gx // max(wx, 1), gy, gz, wx, lds, vg)
    agg[k][0] += 1; agg[k][1] += dur / 1e3
short = lambda k: re.sub(r"\(.*", "", k.replace("void saunet::", "").replace("saunet::", "").replace("unsigned short", "bf16"))[:70]
rows = sorted(((v[1] / steps, v[0] / steps, v[1] / v[0], k) for k, v in agg.items()), reverse=True)
tot = sum(r[0] for r in rows)
print("# total %.1f us/step over %d (kernel, geometry) groups; groups below %.0f us/step omitted" % (tot, len(rows), floor))
print("%-9s %-8s %-9s %-18s %-6s %-7s %-5s %s" % ("us/step", "calls/st", "avg_us", "blocks(x,y,z)", "wg", "lds", "vgpr", "kernel"))
for us, calls, avg, (name, bx, gy, gz, wx, lds, vg) in rows:
    if us < floor:
        continue
    print("%-9.1f %-8.1f %-9.2f %-18s %-6d %-7d %-5d %s" % (us, calls, avg, "%d,%d,%d" % (bx, gy, gz), wx, lds, vg, short(name)))
