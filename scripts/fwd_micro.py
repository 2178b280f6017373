"""Forward 3x3 conv kernels at the step's geometries (B=32, 256x256 input): us per launch.  python scripts/fwd_micro.py"""
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


def case(n, cin, h, cout, k, name, pro=True, stats=True, check=False):
    dt = torch.bfloat16
    x = torch.randn(n, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
    sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.1
    out = HF.new_act(n, cout, h, h, dt, "cuda")
    st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    f = lambda: HF.conv_forward_raw(x, w, None, 1, k // 2, pro=(sc, sh, True) if pro else None, out=out, stats=st if stats else None)
    us = bench(f)
    P = n * h * h
    fl = 2.0 * P * cin * k * k * cout
    by = P * (cin + cout) * 2.0
    extra = ""
    if check:
        st.zero_(); f(); torch.cuda.synchronize()
        a = torch.relu(x.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float() if pro else x.float()
        ref = torch.nn.functional.conv2d(a[:2], w.detach().to(dt).float(), padding=k // 2)
        err = float((out[:2].float() - ref).abs().max() / ref.abs().max())
        ssum = float((st.sum(0)[0].float() - out.float().sum((0, 2, 3))).abs().max() / out.float().sum((0, 2, 3)).abs().max())
        extra = "  err %.1e stat %.1e" % (err, ssum)
    print("%-40s %8.1f us  %6.1f TF/s  %6.0f GB/s (algorithmic)%s" % (name, us, fl / us / 1e6, by / us / 1e3, extra), flush=True)


B = 32
for blk, h in enumerate((128, 64, 32, 16), 1):
    case(B, 128, h, 32, 3, "block%d conv2 fwd 3x3 128->32 @%d" % (blk, h), check=True)
case(B, 64, 256, 64, 3, "res1 conv 3x3 64->64 @256", pro=False, check=True)
case(B, 32, 256, 32, 3, "res2 conv 3x3 32->32 @256", pro=False, check=True)
case(B, 16, 256, 16, 3, "res3 conv 3x3 16->16 @256", pro=False, check=True)
case(B, 64, 256, 32, 3, "dec0 conv 3x3 64->32 @256", pro=False)
case(B, 64, 128, 48, 3, "dec1 conv 3x3 64->48 @128", pro=False)
case(B, 256, 128, 64, 3, "dec2 c3x3rb 256->64 @128", pro=False)
case(B, 512, 64, 128, 3, "dec3 c3x3rb 512->128 @64", pro=False)
case(B, 1024, 32, 256, 3, "dec4 c3x3rb 1024->256 @32", pro=False)
case(B, 1536, 16, 512, 3, "dec5 c3x3rb 1536->512 @16", pro=False)
for blk, (h, cin) in enumerate(((128, 160), (64, 320), (32, 640), (16, 768)), 1):
    case(B, cin, h, 128, 1, "block%d conv1 fwd 1x1 %d->128 @%d" % (blk, cin, h))
