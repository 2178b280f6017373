#!/bin/bash
# MFMA-pipe utilisation of the MFMA-bound forward convolutions, one geometry per rocprofv3 counter pass (GPU box, repo root):
#   scripts/mfma_table.sh  -> gpurun_out/mfma_table.txt
# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), as in scripts/step_pmc_summary.py.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mfma_tab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in center dec5 dec4 dec3 dec2 mrfup5 mrfup3 res1; do
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/$c -o p -- python $R/scripts/one_kernel.py $c 6 > $O/$c.log 2>&1
done
python - <<PY > $R/gpurun_out/mfma_table.txt
import sqlite3, collections, glob
FL = {"center": 2*32*8*8*1024*9*512, "dec5": 2*32*16*16*1536*9*512, "dec4": 2*32*32*32*1024*9*256, "dec3": 2*32*64*64*512*9*128, "dec2": 2*32*128*128*256*9*64,
      "mrfup5": 2*32*16*16*512*4*512, "mrfup3": 2*32*64*64*128*4*128, "res1": 2*32*256*256*64*9*64}
print("# MFMA-pipe utilisation per forward geometry (B=32, bf16; 6 launches each, averages; scripts/mfma_table.sh)")
print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024);  TF/s from the kernel-trace duration;  LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print("%-8s %-62s %9s %8s %9s %8s" % ("case", "kernel", "us", "TF/s", "MFMA busy", "LDS cfl"))
for c in ["center", "dec5", "dec4", "dec3", "dec2", "mrfup5", "mrfup3", "res1"]:
    try:
        db = glob.glob("$O/%s/**/p_results.db" % c, recursive=True) + glob.glob("$O/%s/p_results.db" % c)
        con = sqlite3.connect(db[0])
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.defaultdict(int)
        for k, cn, v, n, d in con.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name"):
            if "saunet" not in k or "pack" in k: continue
            kk = k.split("(")[0].replace("saunet::", "").replace("unsigned short", "bf16")[-62:]
            acc[kk][cn] = v / n; dur[kk] = d / n
        tot = sum(dur.values())
        for kk in acc:
            a = acc[kk]
            busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(a.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
            cfl = a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)
            print("%-8s %-62s %9.1f %8.1f %8.1f%% %7.1f%%" % (c, kk, dur[kk] / 1e3, FL[c] / max(tot, 1) / 1e3 if dur[kk] == max(dur.values()) else 0.0, 100 * busy, 100 * cfl))
    except Exception as e:
        print(c, "failed:", e)
PY
rm -rf $O
cat $R/gpurun_out/mfma_table.txt
