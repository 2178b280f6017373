cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for b in 3 4; do
for k in 0 1; do
echo "wide_k=$k"
SAUNET_IGEMM_WIDE_K=$k python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids
done
done
python -m pytest tests/test_hip_dense.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -3
