#!/bin/bash
# MFMA-pipe utilisation of the LDS-DMA 3x3 kernel per variant (SAUNET_MM_VAR) and geometry; GPU box, repo root.
#   scripts/mm_pmc.sh "0 1 2" "dec3 dec5"   -> gpurun_out/mm_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${1:-0}; CASES=${2:-"dec5 dec4 dec3 dec2"}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do for c in $CASES; do
  rm -rf /tmp/p_$c
  SAUNET_MM_VAR=$v timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p_$c -o p -- python $R/scripts/one_kernel.py $c 6 > /tmp/p_$c.log 2>&1
  python - <<PY
import sqlite3, glob, collections
db = glob.glob("/tmp/p_$c/**/p_results.db", recursive=True)
con = sqlite3.connect(db[0])
acc = collections.defaultdict(dict)
for k, cn, v, n, d in con.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"):
    if "conv3x3" in k or "conv_igemm" in k: acc[k.split("(")[0][-40:]][cn] = v / n; acc[k.split("(")[0][-40:]]["dur_us"] = d / n / 1e3
for k, a in acc.items():
    wc = max(a.get("SQ_WAVE_CYCLES", 1), 1)
    print("var $v %-6s %-40s %7.1f us  mfma busy %5.1f%%  clk %.2f GHz  wait_any %4.1f%%  wait_inst %4.1f%% (lds %4.1f%%)  lds busy %4.1f%% cfl %4.1f%%" % ("$c", k, a["dur_us"],
          100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (a.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024), a.get("GRBM_GUI_ACTIVE", 0) / 8 / a["dur_us"] / 1e3,
          100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_LDS", 0) / wc,
          100 * a.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / (a.get("GRBM_GUI_ACTIVE", 1) / 8), 100 * a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done; done | tee $R/gpurun_out/mm_pmc.txt
