"""float32-storage convolution accuracy against float64 (exact-f32 MFMA vs the 3 x bf16 split): python scripts/f32_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import saunet_amd as S
HF = S.functional
torch.manual_seed(0)
for (n, cin, h, cout, k) in [(2, 512, 32, 128, 3), (2, 1024, 16, 128, 1), (2, 128, 64, 32, 3), (1, 64, 128, 64, 3)]:
    x = torch.randn(n, cin, h, h).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter((torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).cuda())
    y = HF.conv_forward_raw(x, w, None, 1, k // 2)
    ref = F.conv2d(x.double().cpu(), w.detach().double().cpu(), padding=k // 2)
    r32 = F.conv2d(x.cpu(), w.detach().cpu(), padding=k // 2)
    e = (y.double().cpu() - ref)
    e32 = (r32.double() - ref)
    print("conv %dx%d %4d->%-4d @%-3d  HIP: rms err %.3e  max %.3e  mean(signed) %+.3e | torch-cpu f32: rms %.3e max %.3e   (output rms %.3f)" % (
        k, k, cin, cout, h, float(e.pow(2).mean().sqrt()), float(e.abs().max()), float(e.mean()), float(e32.pow(2).mean().sqrt()), float(e32.abs().max()), float(ref.pow(2).mean().sqrt())))
