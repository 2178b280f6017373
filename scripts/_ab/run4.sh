cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for b in 3 4; do
for cfg in "128 512" "192 512" "256 512" "256 1024" "192 1024"; do
set -- $cfg
echo "group_small=$1 blocks_small=$2"
SAUNET_DG_GROUP_SMALL=$1 SAUNET_DG_BLOCKS_SMALL=$2 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids
done
done
