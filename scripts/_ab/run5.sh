cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
for c in conv1dgrad4 conv2dgrad4 conv1fwd4 conv2fwd4 conv1dgrad3 conv2dgrad3 conv1fwd3 conv2fwd3; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done
