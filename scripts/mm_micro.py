"""The MFMA-bound 3x3 convolutions of the step (B=32, 256x256 input, bf16): forward and data gradient, us per launch and TF/s.
python scripts/mm_micro.py [fwd|dgrad|all]"""
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


CASES = [("center", 1024, 8, 512), ("dec5", 1536, 16, 512), ("dec4", 1024, 32, 256), ("dec3", 512, 64, 128), ("dec2", 256, 128, 64),
         ("res1", 64, 256, 64), ("dec0", 64, 256, 32), ("dec1", 64, 128, 48)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
B = 32
dt = torch.bfloat16
for name, cin, h, cout in CASES:
    x = torch.randn(B, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") * 0.03)
    out = HF.new_act(B, cout, h, h, dt, "cuda")
    st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    fl = 2.0 * B * h * h * cin * 9 * cout
    if which in ("fwd", "all"):
        us = bench(lambda: HF.conv_forward_raw(x, w, None, 1, 1, out=out, stats=st))
        print("%-8s fwd   3x3 %4d->%-4d @%-3d  %8.1f us  %7.1f TF/s  (%.3f of 2500)" % (name, cin, cout, h, us, fl / us / 1e6, fl / us / 1e6 / 2500), flush=True)
    if which in ("dgrad", "all"):
        dy = torch.randn(B, cout, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        dx = HF.new_act(B, cin, h, h, dt, "cuda")
        us = bench(lambda: HF.conv_dgrad_raw(dy, w, (B, cin, h, h), 1, 1, out=dx))
        print("%-8s dgrad 3x3 %4d->%-4d @%-3d  %8.1f us  %7.1f TF/s  (%.3f of 2500)" % (name, cout, cin, h, us, fl / us / 1e6, fl / us / 1e6 / 2500), flush=True)
