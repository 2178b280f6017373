"""ConvTranspose2d(k4 s2 p1) weight gradients at the decoder geometries of the bench step (B = 32): us per launch, direct (parity tiles) vs im2col path.
python scripts/convt_wgrad_micro.py"""
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

for name, ci, co, h in (("dec4.mrf.up", 512, 512, 16), ("dec3.mrf.up", 256, 256, 32), ("dec2.mrf.up", 128, 128, 64), ("dec1.convT", 48, 32, 128)):
    dt = torch.bfloat16
    x = torch.randn(32, ci, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, co, 2 * h, 2 * h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(ci, co, 4, 4, device="cuda") * 0.03)
    out = []
    for direct in (True, False):
        HF.CONVT_WGRAD_DIRECT = direct
        out.append(bench(lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 2, 1, transposed=True))))
    by = (x.numel() + dy.numel()) * 2.0
    print("%-12s %4d->%-4d @%-3d  direct %7.1f us (%5.0f GB/s algorithmic)   im2col %7.1f us" % (name, ci, co, h, out[0], by / out[0] / 1e3, out[1]))
