cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for tp in 4096 4097 20000; do for b in 1 2; do echo "MAXTP=$tp"; SAUNET_DG_LDS_MAXTP=$tp python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
