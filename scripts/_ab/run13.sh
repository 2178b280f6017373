cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a
for cfg in "1 1" "1 0" "0 0"; do
set -- $cfg
rm -f gpurun_out/r2_parity_bf16_256.txt
SAUNET_DENSE_BWD_FUSED=$1 SAUNET_TRANSITION_FOLD=$2 python -m pytest tests/test_hip_parity_bf16.py -q -m gpu -k "bf16_gradients_at_256" 2>&1 | tail -1
cp gpurun_out/r2_parity_bf16_256.txt gpurun_out/r05a/parity_fused$1_fold$2.txt
done
