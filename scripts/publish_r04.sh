#!/bin/bash
# Copy the judged summaries of a scripts/collect_r04.sh run (gpurun_out/r04/, merged back by gpurun) into profiles/ (tracked).
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/r04; P=$R/profiles
cp $O/bench.json $P/r04_bench.json; cp $O/bench_infer.json $P/r04_bench_infer.json; cp $O/bench_cfg3_b64.json $P/r04_bench_cfg3_b64.json
cp $O/bench_cfg4_512_f32.json $P/r04_bench_cfg4_512_f32.json; cp $O/bench_eager.json $P/r04_bench_eager.json
cp $O/r04_f_step_kernel_stats.txt $O/r04_roofline_kernel_rocprof.txt $P/
cp $O/step_by_geometry.txt $P/r04_step_by_geometry.txt; cp $O/step_pmc_summary.txt $P/r04_step_pmc_summary.txt
cp $O/step_pmc.json $O/roofline_pmc.json $P/; cp $O/phase_timing.txt $P/r04_phase_timing_raw.txt
cp $O/mfma_table.txt $P/r04_mfma_table.txt
cat $O/micro_fwd.txt $O/micro_mm.txt $O/micro_wgrad.txt $O/micro_convt_wgrad.txt $O/micro_gate.txt | grep -v amdgpu.ids > $P/r04_kernel_microbench.txt
python - <<PY
import json
b = json.load(open("$P/r04_bench.json")); r = b["roofline"]
print("step %.3f ms = %.1f slices/s; roofline %.3f, launch_mix %.4f; step traffic %.1f GB (source file)" % (b["ms_per_step"], b["value"], r["frac"], r["launch_mix"]["frac"], r["step"]["traffic_bytes"] / 1e9))
print("step_pmc.json:", open("$P/step_pmc.json").read()[:200])
PY
