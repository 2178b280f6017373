cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
for b in 3 4; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
for c in conv2dgrad3 conv2dgrad4; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done
