cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_timing.so
for c in k3corr1; do python scripts/phase_timing.py $c 2>&1 | grep -v amdgpu.ids; done
