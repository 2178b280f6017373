cd $GRAFT_REPO_ROOT
SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so python scripts/phase_timing.py k4lds3 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -3
for b in 3 4; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -3; done
