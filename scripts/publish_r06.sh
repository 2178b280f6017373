#!/bin/bash
# Copy the judged summaries of a scripts/collect_r06.sh run (gpurun_out/r06/, merged back by gpurun) into profiles/ (tracked).
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/r06; P=$R/profiles
cp $O/bench.json $P/r06_bench.json; cp $O/bench_infer.json $P/r06_bench_infer.json; cp $O/bench_cfg3_b64.json $P/r06_bench_cfg3_b64.json
cp $O/bench_cfg4_512_f32.json $P/r06_bench_cfg4_512_f32.json; cp $O/bench_eager.json $P/r06_bench_eager.json
cp $O/r06_f_step_kernel_stats.txt $O/r06_roofline_kernel_rocprof.txt $P/
cp $O/step_by_geometry.txt $P/r06_step_by_geometry.txt; cp $O/step_pmc_summary.txt $P/r06_step_pmc_summary.txt
cp $O/step_pmc.json $O/roofline_pmc.json $P/; cp $O/phase_timing.txt $P/r06_phase_timing_raw.txt
cp $O/mfma_table.txt $P/r06_mfma_table.txt
cp $O/dense_chain.txt $P/r06_dense_chain_micro.txt; cp $O/census_table.txt $P/r06_census_table.txt; cp $O/bench_rehearsal_2ranks.json $P/r06_bench_rehearsal_2ranks.json
cat $O/micro_fwd.txt $O/micro_mm.txt $O/micro_wgrad.txt $O/micro_convt_wgrad.txt $O/micro_gate.txt | grep -v amdgpu.ids > $P/r06_kernel_microbench.txt
python - <<PY
import json
b = json.load(open("$P/r06_bench.json")); r = b["roofline"]
print("step %.3f ms = %.1f slices/s; roofline kernel %s frac %.3f; step traffic %.1f GB (source file)" % (b["ms_per_step"], b["value"], r["kernel"][:40], r["frac"], (r["step"]["traffic_bytes"] or 0) / 1e9))
print("step_pmc.json:", open("$P/step_pmc.json").read()[:200])
PY
