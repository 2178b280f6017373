cd $GRAFT_REPO_ROOT
export SAUNET_HIP_LIB=$PWD/scripts/_ab/libsaunet_ab.so
for cfg in "8 64 288 256" "8 64 480 448" "32 32 640 608"; do
SAUNET_DENSE_CONV1_SMALL=0 python scripts/conv1_small_cmp.py /tmp/a.pt $cfg 2>&1 | grep -v amdgpu
SAUNET_DENSE_CONV1_SMALL=1 python scripts/conv1_small_cmp.py /tmp/b.pt $cfg 2>&1 | grep -v amdgpu
python - <<'PY'
import torch
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
for k in a:
    d = (a[k].double() - b[k].double()).abs()
    print(k, "max abs diff %.3e  (scale %.3e)  differing elements %d of %d" % (float(d.max()), float(a[k].double().abs().max()), int((d > 0).sum()), d.numel()))
PY
done
