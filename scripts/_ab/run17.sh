cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -4
for b in 3 4; do python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids; done
