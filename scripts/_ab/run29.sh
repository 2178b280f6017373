cd $GRAFT_REPO_ROOT
for pr in 1 0; do for b in 2 3 4; do echo "PAIRS=$pr"; SAUNET_DENSE_BWD_PAIRS=$pr python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids | tail -1; done; done
