#!/bin/bash
# Run on the GPU box from the repo root (gpurun): collects the rocprofv3 evidence that profiles/ is built from.
#   1. kernel-trace + stats of the roofline probe (python bench.py --roofline-only)
#   2. PMC passes of the same command: FETCH_SIZE, then WRITE_SIZE (+ L2 hit/miss)  [separate passes, no trace domains]
#   3. the same two PMC passes on scripts/probes/rowpiece_probe (known byte counts) to calibrate the counters for this access pattern
#   4. kernel-trace + stats of the eager training step
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round_prof
rm -rf $O; mkdir -p $O
[ -x $R/scripts/probes/rowpiece_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/probes/rowpiece_probe.hip -o $R/scripts/probes/rowpiece_probe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/rf_stats -o p -- python $R/bench.py --roofline-only --no-launch-mix > $O/rf_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/rf_fetch -o p -- python $R/bench.py --roofline-only --no-launch-mix > $O/rf_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/rf_write -o p -- python $R/bench.py --roofline-only --no-launch-mix > $O/rf_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/rf_sq -o p -- python $R/bench.py --roofline-only --no-launch-mix > $O/rf_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/cal_fetch -o p -- $R/scripts/probes/rowpiece_probe > $O/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/cal_write -o p -- $R/scripts/probes/rowpiece_probe > $O/cal_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/step -o p -- python $R/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $O/step.log 2>&1
grep -h '"roofline"\|"metric"' $O/*.log | cut -c1-400
ls $O
