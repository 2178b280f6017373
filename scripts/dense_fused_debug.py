"""per-tensor deviations of the fused (two launches per layer) dense-block backward from the four-launch one and from float64"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import saunet_amd as S
from test_hip_dense import ref_block, rel_l2
HF = S.functional
layers, cin, shape = int(sys.argv[1]), int(sys.argv[2]), tuple(int(v) for v in sys.argv[3:6])
torch.manual_seed(17 + layers)
n, h, w = shape
dtype = torch.bfloat16
block = S.modules._DenseBlock(layers, cin).cuda().train()
with torch.no_grad():
    for m in block.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(4.0, 6.0)
x0 = torch.randn(n, cin, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
cot, grads = None, {}
for fused in (True, False):
    HF.DENSE_BWD_FUSED = fused
    block.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    y = block(x)
    if cot is None:
        cot = torch.randn(y.shape, device="cuda").to(dtype)
    (y.float() * cot.float()).sum().backward()
    grads[fused] = {"x": x.grad.float().clone(), **{k: v.grad.float().clone() for k, v in block.named_parameters()}}
ry, xr, prm = ref_block(block, x0)
(ry * cot.double()).sum().backward()
ref = {"x": xr.grad, **{k: prm[k].grad for k in prm}}
for k in grads[True]:
    print("%-32s fused-vs-unfused %.3e   fused-vs-f64 %.3e   unfused-vs-f64 %.3e" % (k, rel_l2(grads[True][k], grads[False][k]), rel_l2(grads[True][k], ref[k]), rel_l2(grads[False][k], ref[k])))
