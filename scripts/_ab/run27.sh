cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for nb in 512 1024 256; do echo "CW_BLOCKS=$nb"; SAUNET_DGRAD3_CW_BLOCKS=$nb python scripts/dense_chain_micro.py 3 2>&1 | grep -v amdgpu.ids | tail -1; done
echo "CW off"; SAUNET_DGRAD3_CW=0 python scripts/dense_chain_micro.py 3 2>&1 | grep -v amdgpu.ids | tail -1
