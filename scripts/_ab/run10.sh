cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py -x -q -m gpu -k "small_map or fused" 2>&1 | tail -3
for b in 3 4; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids; done
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
for c in conv1small4 conv1small3; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done
