import sys, os
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
    return e0.elapsed_time(e1) / reps

def conv_case(n, cin, h, cout, k, dtype, pro, stats, name):
    x = torch.randn(n, cin, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
    scale = torch.rand(cin, device="cuda") + 0.5; shift = torch.randn(cin, device="cuda") * 0.1
    out = HF.new_act(n, cout, h, h, dtype, "cuda")
    st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    f = lambda: HF.conv_forward_raw(x, w, None, 1, k // 2, pro=(scale, shift, True) if pro else None, out=out, stats=st if stats else None)
    ms = bench(f)
    fl = 2.0 * n * h * h * cin * k * k * cout
    print("%-34s pro=%d stats=%d  %.3f ms  %.1f TF/s" % (name, pro, stats, ms, fl / ms / 1e9))

dt = torch.bfloat16
for pro, st in ((0, 0), (1, 0), (0, 1), (1, 1)):
    conv_case(32, 128, 128, 32, 3, dt, pro, st, "3x3 128->32 @128 B32")
for pro, st in ((0, 0), (1, 1)):
    conv_case(32, 256, 128, 128, 1, dt, pro, st, "1x1 256->128 @128 B32")
    conv_case(32, 64, 256, 64, 3, dt, pro, st, "3x3 64->64 @256 B32 (res1)")
    conv_case(32, 1024, 32, 256, 3, dt, pro, st, "3x3 1024->256 @32 B32 (dec4)")
conv_case(32, 128, 128, 32, 3, torch.float32, 1, 1, "f32 3x3 128->32 @128 B32")
# wgrad
def wgrad_case(n, cin, h, cout, k, dtype, name):
    x = torch.randn(n, cin, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, cout, h, h, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
    f = lambda: HF.conv_wgrad_raw(x, dy, w, 1, k // 2)
    ms = bench(f)
    fl = 2.0 * n * h * h * cin * k * k * cout
    print("%-34s wgrad  %.3f ms  %.1f TF/s" % (name, ms, fl / ms / 1e9))
wgrad_case(32, 128, 128, 32, 3, dt, "3x3 128->32 @128 B32")
wgrad_case(32, 256, 128, 128, 1, dt, "1x1 256->128 @128 B32")
wgrad_case(32, 64, 256, 64, 3, dt, "3x3 64->64 @256 B32 (res1)")
wgrad_case(32, 1024, 32, 256, 3, dt, "3x3 1024->256 @32 B32 (dec4)")
print("--- 1x1 split")
for pro, st in ((0, 0), (1, 0), (0, 1), (1, 1)):
    conv_case(32, 256, 128, 128, 1, dt, pro, st, "1x1 256->128 @128 B32")
for pro, st in ((0, 0), (0, 1)):
    conv_case(32, 64, 256, 64, 3, dt, pro, st, "3x3 64->64 @256 B32 (res1)")
