cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -2
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for h2 in 1 0; do for b in 1 2; do echo "HALO2=$h2"; SAUNET_DGRAD3_HALO2=$h2 timeout 300 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
python scripts/phase_timing.py k3corr1 2>&1 | grep -v amdgpu.ids
