"""Turn the rocprofv3 databases written by scripts/collect_profiles.sh (gpurun_out/round_prof/) into the committed
profiles/ artefacts: python scripts/make_profiles.py [tag]   (tag defaults to r01)"""
import json, os, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "round_prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

def pmc(name, like):
    con = sqlite3.connect("%s/%s/p_results.db" % (O, name))
    return {r[0]: (r[1], r[2]) for r in con.execute(
        "select counter_name, sum(value)/count(*), count(*) from counters_collection where kernel_name like ? group by counter_name", (like,))}

f, w, sq = pmc("rf_fetch", "%dense_dgrad_kernel%"), pmc("rf_write", "%dense_dgrad_kernel%"), pmc("rf_sq", "%dense_dgrad_kernel%")
cf = {m: pmc("cal_fetch", "%%k<%d>%%" % m)["FETCH_SIZE"][0] for m in (0, 1, 2)}
cw = {m: pmc("cal_write", "%%k<%d>%%" % m)["WRITE_SIZE"][0] for m in (0, 1, 2)}
con = sqlite3.connect(O + "/rf_stats/p_results.db")
avg = [r for r in con.execute("select name,total_calls,average from top_kernels where name like '%dense_dgrad_kernel%'")][0]
line = json.loads([l for l in open(O + "/rf_stats.log") if '"roofline"' in l][0])["roofline"]
fetch_b, write_b = f["FETCH_SIZE"][0] * 1024 * 2, w["WRITE_SIZE"][0] * 1024
rec = {"kernel": line["kernel"],
       "command": "python bench.py --roofline-only  (rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group: scripts/collect_profiles.sh)",
       "launches_sampled": int(f["FETCH_SIZE"][1]),
       "FETCH_SIZE_KB_per_launch": round(f["FETCH_SIZE"][0], 1), "WRITE_SIZE_KB_per_launch": round(w["WRITE_SIZE"][0], 1),
       "correction": "FETCH_SIZE x2 (gfx950: 128-byte read requests tallied at 64 B; MI355X_MICROARCH.md, HBM section), WRITE_SIZE x1",
       "calibration": {"probe": "scripts/probes/rowpiece_probe (known bytes per launch: 393216 KB read, 196608 KB written)",
                       "FETCH_SIZE_KB": {"coalesced": round(cf[0], 1), "8B_row_pieces": round(cf[1], 1), "16B_row_pieces": round(cf[2], 1)},
                       "WRITE_SIZE_KB": {"coalesced": round(cw[0], 1), "8B_row_pieces": round(cw[1], 1), "16B_row_pieces": round(cw[2], 1)}},
       "fetch_bytes_per_launch": int(fetch_b), "write_bytes_per_launch": int(write_b),
       "traffic_bytes_per_launch": int(fetch_b + write_b),
       "algorithmic_bytes_per_launch": line["algorithmic_bytes"],
       "traffic_over_algorithmic": round((fetch_b + write_b) / line["algorithmic_bytes"], 3),
       "TCC_hit_rate": round(w["TCC_HIT_sum"][0] / (w["TCC_HIT_sum"][0] + w["TCC_MISS_sum"][0]), 3),
       "kernel_trace_avg_us": round(avg[2], 2), "kernel_trace_calls": avg[1], "bench_hip_event_ms": line["ms"],
       "SQ": {k: round(v[0]) for k, v in sq.items()}}
json.dump(rec, open(os.path.join(ROOT, "profiles", "roofline_pmc.json"), "w"), indent=1)

def summary(db, steps, families=False):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prof_summary.py"), db, str(steps)] + (["--families"] if families else []),
                          capture_output=True, text=True).stdout

rf = summary(O + "/rf_stats/p_results.db", 1).split("\n")
rf = "\n".join(l[:230] for l in rf[1:10])
open(os.path.join(ROOT, "profiles", "%s_roofline_kernel_rocprof.txt" % tag), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --roofline-only   (scripts/collect_profiles.sh)\n"
    "# the probe launches the dominant kernel 33 times (3 warm-up + 30 timed with HIP events on the launch stream)\n" + rf +
    "\n# bench.py line of the same run:\n" + json.dumps({"roofline": line}) + "\n")
open(os.path.join(ROOT, "profiles", "%s_f_step_kernel_stats.txt" % tag), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline\n"
    "# (eager mode so that every launch is attributed; 15 steps incl. warm-up; per-step = total / 15)\n" + summary(O + "/step/p_results.db", 15, families=True))
print(json.dumps(rec)[:600])
