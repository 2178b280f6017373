cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dense.py -x -q -m gpu -k "pairs or fused" 2>&1 | tail -3
for b in 2 3; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
for c in k4pair3; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done
