#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/prof_step.sh <name> [bench args]   -> gpurun_out/<name>/p_results.db
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/$name
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$name -o p -- python $R/bench.py --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras "$@" > $R/gpurun_out/$name.log 2>&1
grep '"metric"' $R/gpurun_out/$name.log | cut -c1-200
