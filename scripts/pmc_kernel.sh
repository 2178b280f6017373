#!/bin/bash
# usage: scripts/pmc_kernel.sh <case> <out-name>  -> gpurun_out/<out-name>.txt : SQ / LDS / MFMA / HBM counters of one conv geometry
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$1; N=$2
O=$R/gpurun_out/pmc_$N; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/scripts/one_kernel.py $C 6"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/a -o p -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $O/b -o p -- $B > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c -o p -- $B > $O/c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/d -o p -- $B > $O/d.log 2>&1
python - <<PY > $R/gpurun_out/$N.txt
import sqlite3, collections
out = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
for p in "abcd":
    try:
        con = sqlite3.connect("$O/%s/p_results.db" % p)
        for k, c, v, n, d in con.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"):
            if "saunet" in k and ("conv" in k or "dense" in k):
                out[k.split("(")[0][-70:]][c] = v / n; dur[k.split("(")[0][-70:]] = d / n
    except Exception as e:
        print("pass", p, "failed", e)
for k, cs in out.items():
    print(k, " avg duration %.1f us" % (dur[k] / 1e3))
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f" % (c, v))
PY
rm -rf $O
cat $R/gpurun_out/$N.txt
