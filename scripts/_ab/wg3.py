import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import saunet_amd as S
HF = S.functional
dt = torch.bfloat16
for (cin, h, cout) in ((512, 64, 128), (256, 128, 64), (1024, 32, 256)):
    x = torch.randn(32, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, cout, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") * 0.03)
    for _ in range(5):
        HF.GRADS.reset(); HF.conv_wgrad_raw(x, dy, w, 1, 1)
torch.cuda.synchronize()
