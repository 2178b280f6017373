cd $GRAFT_REPO_ROOT
for pr in 1 0 1 0; do echo "PAIRS=$pr"; SAUNET_DENSE_BWD_PAIRS=$pr python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | cut -c1-220; done
