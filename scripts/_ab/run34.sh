cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for tp in 4097 20000; do for pr in 1 0; do echo "MAXTP=$tp PAIRS=$pr"; SAUNET_DG_LDS_MAXTP=$tp SAUNET_DENSE_BWD_PAIRS=$pr timeout 300 python scripts/dense_chain_micro.py 1 2>&1 | grep -v amdgpu.ids | tail -1; done; done
