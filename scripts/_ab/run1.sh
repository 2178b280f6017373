cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
python -m pytest tests/test_hip_dense.py tests/test_hip_conv_mm.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05a/tests.txt
cat gpurun_out/r05a/tests.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*' | tee gpurun_out/r05a/bench_fused.txt
SAUNET_DENSE_BWD_FUSED=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*' | tee gpurun_out/r05a/bench_unfused.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*' | tee -a gpurun_out/r05a/bench_fused.txt
SAUNET_DENSE_BWD_FUSED=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*' | tee -a gpurun_out/r05a/bench_unfused.txt
