cd $GRAFT_REPO_ROOT
for lib in ab ab8; do
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_$lib.so
for bm in 64 128; do for b in 4; do echo "lib=$lib BM=$bm"; SAUNET_DG_LDS_BM=$bm python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
done
