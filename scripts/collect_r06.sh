#!/bin/bash
# Round-6 evidence run (GPU box, from the repo root).  Everything lands in gpurun_out/r06/ ; copy what is judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
# 1. profiles first: bench.py reads the kernel ranking and the per-kernel PMC traffic of THIS round from profiles/
bash scripts/collect_profiles.sh > $O/collect_profiles.log 2>&1
python scripts/make_profiles.py r06 > $O/make_profiles.log 2>&1
python scripts/prof_by_geometry.py gpurun_out/round_prof/step/p_results.db 15 25 > $O/step_by_geometry.txt
bash scripts/collect_step_pmc.sh > $O/step_pmc.log 2>&1
cp gpurun_out/step_pmc.json profiles/step_pmc.json
# 2. the bench lines and micro-benchmarks
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python bench.py --infer --steps 30 > $O/bench_infer.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --batch 64 --no-cpu-baseline --no-roofline --no-extras > $O/bench_cfg3_b64.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --size 512 --batch 8 --dtype f32 --no-cpu-baseline --no-roofline --no-extras > $O/bench_cfg4_512_f32.json 2>> $O/bench.err
python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline --no-extras > $O/bench_eager.json 2>> $O/bench.err
python scripts/fwd_micro.py > $O/micro_fwd.txt 2>&1
python scripts/wgrad_micro.py > $O/micro_wgrad.txt 2>&1
python scripts/mm_micro.py all > $O/micro_mm.txt 2>&1
python scripts/convt_wgrad_micro.py > $O/micro_convt_wgrad.txt 2>&1
python scripts/gate_micro.py > $O/micro_gate.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/gatep -o g -- python $R/scripts/gate_micro.py 5 > /dev/null 2>&1; python $R/scripts/prof_summary.py /tmp/gatep/g_results.db 1 | grep -E "gate|calls" ) >> $O/micro_gate.txt 2>&1
( export SAUNET_HIP_LIB=scripts/_ab/libsaunet_timing.so; for c in conv2fwd conv2wgrad conv1wgrad dec3wgrad conv1dgrad conv1dgrad3 conv2dgrad conv2dgrad3 dec3mm dec5mm conv1fwd conv1fwd3 conv1small3 conv1small4 conv1dgrad3 conv1dgrad4 conv2dgrad3 conv2dgrad4 conv2fwd3 conv2fwd4 k4lds3 k4lds4 k4pair3 k4pair2; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done ) > $O/phase_timing.txt
for b in 1 2 3 4; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids; ( echo -n "[layer pairs off] "; SAUNET_DENSE_BWD_PAIRS=0 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1 ); SAUNET_DENSE_BWD_FUSED=0 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids; done > $O/dense_chain.txt
python scripts/census_table.py 2>&1 | grep -v amdgpu.ids > $O/census_table.txt
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --no-cpu-baseline 2>> $O/bench.err | grep "^{" > $O/bench_rehearsal_2ranks.json
bash scripts/mfma_table.sh > $O/mfma_table.log 2>&1; cp gpurun_out/mfma_table.txt $O/mfma_table.txt
cp gpurun_out/step_pmc_summary.txt $O/step_pmc_summary.txt; cp gpurun_out/step_pmc.json $O/step_pmc.json
cp profiles/roofline_pmc.json $O/roofline_pmc.json; cp profiles/r06_roofline_kernel_rocprof.txt profiles/r06_f_step_kernel_stats.txt $O/ 2>/dev/null
rm -rf gpurun_out/round_prof
tail -c 600 $O/bench.json; echo; tail -3 $O/step_pmc_summary.txt
