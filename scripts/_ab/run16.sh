cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in scripts/_ab/libsaunet_base.so shape-attentive-unet_amd/libsaunet_hip.so; do
echo -n "$(basename $L) f32 "
SAUNET_HIP_LIB=$PWD/$L python bench.py --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*'
done; done
python -m pytest tests/test_hip_dense.py tests/test_hip_ops.py -q -m gpu 2>&1 | tail -2
