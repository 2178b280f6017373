R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for b in 2 3; do
rocprofv3 --kernel-trace --stats -d /tmp/prof$b -o p -- python $R/scripts/dense_chain_micro.py $b > /dev/null 2>&1
python $R/scripts/prof_summary.py /tmp/prof$b/p_results.db 1 2>&1 | cut -c1-150 | grep -v "pack_weight\|at::native" | head -11
done
