cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_hip_saunet.py tests/test_hip_parity_bf16.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
SAUNET_DENSE_BWD_FUSED=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
