cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for v in 0 1 0 1; do
SAUNET_DENSE_CONV1_SMALL=$v python -m pytest tests/test_hip_parity_bf16.py -q -m gpu -k "bf16_gradients_at_256" 2>&1 | tail -1
echo "conv1_small=$v: $(grep -E '^center|^denseblock4|^denseblock3|^dec5' gpurun_out/r2_parity_bf16_256.txt | cut -c1-50 | tr '\n' ';')"
done
