#!/bin/bash
# Whole-step PMC passes (eager mode, 3 warm-up + 4 steps): HBM bytes and MFMA busy cycles per kernel.  One pass per counter group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/step_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-graph --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-extras"      # 2 extra eager steps run before the warm-up: 9 per pass
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/mfma -o p -- $B > $O/mfma.log 2>&1
ls -la $O/*/p_results.db
python $R/scripts/step_pmc_summary.py $O 9 $R/gpurun_out/step_pmc.json > $R/gpurun_out/step_pmc_summary.txt 2>&1
rm -rf $O
tail -3 $R/gpurun_out/step_pmc_summary.txt
