import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_hip_dense import ref_block
import saunet_amd as S

def l2(a, b):
    b = b.double(); a = a.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30)), float((a - b).abs().max() / b.abs().max())

for layers, cin, shape in [(3, 64, (2, 32, 32)), (2, 96, (1, 16, 48)), (4, 40, (3, 16, 16)), (6, 64, (4, 64, 64))]:
    torch.manual_seed(layers * 100 + cin)
    n, h, w = shape
    block = S.modules._DenseBlock(layers, cin).cuda().train()
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    x = torch.randn(n, cin, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = block(x)
    cot = torch.randn(y.shape, device="cuda").to(torch.bfloat16)
    (y.float() * cot.float()).sum().backward()
    ry, xr, prm = ref_block(block, x)
    (ry * cot.double()).sum().backward()
    print(layers, cin, shape, "y", l2(y, ry), "dx", l2(x.grad, xr.grad))
    worst = max((l2(v.grad, prm[k].grad)[0], k) for k, v in block.named_parameters())
    print("   worst param grad (L2 rel):", worst)
