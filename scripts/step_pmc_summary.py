"""Per-kernel HBM traffic and MFMA utilisation of the training step from the PMC passes of scripts/collect_step_pmc.sh.
python scripts/step_pmc_summary.py <dir with fetch/ write/ mfma/> <steps incl. warm-up>  ->  text table on stdout"""
import collections, re, sqlite3, sys
O, steps = sys.argv[1], float(sys.argv[2])

def per_kernel(name, counters):
    con = sqlite3.connect("%s/%s/p_results.db" % (O, name))
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter(); dur = collections.Counter()
    for k, c, v, n, d in con.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"):
        out[k][c] += v
        cnt[k] = n; dur[k] = d
    return out, cnt, dur

f, nf, df = per_kernel("fetch", ["FETCH_SIZE"])
w, nw, dw = per_kernel("write", ["WRITE_SIZE"])
m, nm, dm = per_kernel("mfma", [])
short = lambda k: re.sub(r"\(.*", "", k.replace("void saunet::", "").replace("saunet::", "").replace("unsigned short", "bf16"))[:64]
rows = []
for k in f:
    fb = f[k].get("FETCH_SIZE", 0.0) * 1024 * 2          # gfx950 correction (MI355X_MICROARCH.md): FETCH_SIZE counts half the bytes
    wb = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    d = df[k]                                              # ns, summed over the dispatches of the fetch pass
    mf = m.get(k, {})
    gui = mf.get("GRBM_GUI_ACTIVE", 0.0) / 8.0             # summed over the 8 XCDs
    util = mf.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024) if gui else 0.0   # busy cycles / (cycles x 1024 SIMDs)
    valu = mf.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (gui * 1024) if gui else 0.0
    rows.append((d, k, nf[k], fb, wb, util, valu))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("# whole-step PMC summary: eager training step, B=32 256x256 bf16 (scripts/collect_step_pmc.sh; %g steps incl. warm-up)" % steps)
print("# HBM bytes = FETCH_SIZE*2 + WRITE_SIZE (KB -> bytes; x2 = gfx950 read-request correction); GB/s = bytes / summed kernel time;")
print("# MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); VALU = SQ_ACTIVE_INST_VALU*4 / same")
print("%-8s %-9s %-9s %-10s %-8s %-7s %-7s %s" % ("calls/st", "ms/step", "MB/call", "GB/s", "time%", "MFMA%", "VALU%", "kernel"))
for d, k, n, fb, wb, util, valu in rows[:40]:
    print("%-8.1f %-9.3f %-9.1f %-10.0f %-8.2f %-7.1f %-7.1f %s" % (n / steps, d / steps / 1e6, (fb + wb) / n / 1e6, (fb + wb) / d if d else 0, 100 * d / tot, 100 * util, 100 * valu, short(k)))
print("# all kernels: %.2f ms/step, %.1f GB/step HBM traffic, average %.0f GB/s" % (tot / steps / 1e6, sum(r[3] + r[4] for r in rows) / steps / 1e9, sum(r[3] + r[4] for r in rows) / tot))
if len(sys.argv) > 3:      # machine-readable record for bench.py's roofline.step (profiles/step_pmc.json)
    import json
    per_kernel = {k.replace("void saunet::", "").replace("saunet::", ""): {"launches_per_step": round(n / steps, 2), "traffic_bytes_per_step": (fb + wb) / steps,
                                                                              "fetch_bytes_per_step": fb / steps, "write_bytes_per_step": wb / steps, "ms_per_step": d / steps / 1e6}
                  for d, k, n, fb, wb, util, valu in rows}
    json.dump({"config": [256, 32, "bf16"], "traffic_bytes_per_step": sum(r[3] + r[4] for r in rows) / steps, "per_kernel": per_kernel,
               "kernel_ms_per_step": tot / steps / 1e6, "steps": steps,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph` (scripts/collect_step_pmc.sh); bytes = FETCH_SIZE*2 + WRITE_SIZE"},
              open(sys.argv[3], "w"), indent=1)
