cd $GRAFT_REPO_ROOT
for b in 3 4 2 1; do
python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids
SAUNET_DENSE_BWD_FUSED=0 python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids
done
