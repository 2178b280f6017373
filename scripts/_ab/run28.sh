cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dense.py -x -q -m gpu 2>&1 | tail -15
