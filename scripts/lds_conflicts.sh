#!/bin/bash
# LDS bank-conflict share per kernel over one eager training step (GPU box, repo root) -> gpurun_out/lds_conflicts.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ldsc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O -o p -- python $R/bench.py --no-graph --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $O.log 2>&1
python - <<PY > $R/gpurun_out/lds_conflicts.txt
import sqlite3, collections, glob
db = glob.glob("$O/**/p_results.db", recursive=True) + glob.glob("$O/p_results.db")
con = sqlite3.connect(db[0])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.defaultdict(int)
for k, cn, v, n, d in con.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"):
    kk = k.split("(")[0].replace("saunet::", "").replace("unsigned short", "bf16")[-70:]
    acc[kk][cn] = v; dur[kk] = d; cnt[kk] = n
print("# LDS bank conflicts per kernel over 6 eager training steps (B=32 256x256 bf16): conflict cycles / LDS-array cycles; scripts/lds_conflicts.sh")
print("%-72s %8s %10s %9s" % ("kernel", "calls", "total ms", "conflict"))
for kk in sorted(acc, key=lambda k: -dur[k])[:40]:
    a = acc[kk]
    if a.get("SQ_LDS_IDX_ACTIVE", 0) <= 0: continue
    print("%-72s %8d %10.2f %8.1f%%" % (kk, cnt[kk], dur[kk] / 1e6, 100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf $O
cat $R/gpurun_out/lds_conflicts.txt
