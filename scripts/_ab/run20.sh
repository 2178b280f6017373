cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -3
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for bm in 64 128; do for b in 3 4; do echo "BM=$bm"; SAUNET_DG_LDS_BM=$bm python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
