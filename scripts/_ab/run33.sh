cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -2
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for mt in 384 1025 4097; do for b in 1 2; do echo "MAXTILES=$mt"; SAUNET_DENSE_CONV1_MAXTILES=$mt timeout 300 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
