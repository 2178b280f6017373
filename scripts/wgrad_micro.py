"""Weight-gradient kernels at the DenseNet layer geometries of the bench step (B=32, 256x256 input): us per launch, TF/s, GB/s.
python scripts/wgrad_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional


def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(n, cin, h, cout, k, name, pro=True):
    dt = torch.bfloat16
    x = torch.randn(n, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.1
    f = lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 1, k // 2, pro=(sc, sh, True) if pro else None))
    us = bench(f)
    P = n * h * h
    fl = 2.0 * P * cin * k * k * cout
    by = P * (cin + cout) * 2.0
    print("%-40s %8.1f us  %6.1f TF/s  %6.0f GB/s (algorithmic)" % (name, us, fl / us / 1e6, by / us / 1e3), flush=True)


B = 32
for blk, (h, c0, L) in enumerate([(128, 64, 6), (64, 128, 12), (32, 256, 24), (16, 512, 16)], 1):
    case(B, 128, h, 32, 3, "block%d conv2 wgrad 3x3 128->32 @%d" % (blk, h))
    for l in (0, L // 2, L - 1):
        case(B, c0 + 32 * l, h, 128, 1, "block%d conv1 wgrad 1x1 %d->128 @%d" % (blk, c0 + 32 * l, h))
case(B, 64, 256, 64, 3, "res1 wgrad 3x3 64->64 @256", pro=False)
case(B, 1024, 32, 256, 3, "dec4 wgrad 3x3 1024->256 @32", pro=False)
case(B, 512, 64, 128, 3, "dec3 wgrad 3x3 512->128 @64", pro=False)
case(B, 256, 128, 64, 3, "dec2 wgrad 3x3 256->64 @128", pro=False)
case(B, 64, 256, 64, 3, "res1.conv2 wgrad 3x3 64->64 @256 (prologue)", pro=True)
case(B, 32, 256, 32, 3, "res2 wgrad 3x3 32->32 @256", pro=False)
case(B, 16, 256, 16, 3, "res3 wgrad 3x3 16->16 @256", pro=False)
case(B, 64, 256, 32, 3, "dec0 wgrad 3x3 64->32 @256", pro=False)
case(B, 64, 128, 48, 3, "dec1 wgrad 3x3 64->48 @128", pro=False)
