cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for b in 3 4; do
for y in 0 1; do
echo "yahead=$y"
SAUNET_DG_YAHEAD=$y python scripts/dense_chain_micro.py $b 2>&1 | grep -v amdgpu.ids
done
done
unset SAUNET_HIP_LIB
python -m pytest tests/test_hip_dense.py -x -q -m gpu -k "fused or dense_block" 2>&1 | tail -3
