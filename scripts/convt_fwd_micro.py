"""ConvTranspose2d(4, 2, 1) forward of the decoder's `mrf.up` layers at the bench geometry (B = 32, 256 x 256 input): us per launch,
TFLOP/s, fraction of the dense bf16 peak.  FLOPs = 2 * P_in * 16 * Cin * Cout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
dt = torch.bfloat16
B = 32
for name, c, h in (("dec5 mrf.up 512 @8", 512, 8), ("dec4 mrf.up 512 @16", 512, 16), ("dec3 mrf.up 256 @32", 256, 32), ("dec2 mrf.up 128 @64", 128, 64)):
    x = torch.randn(B, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(c, c, 4, 4, device="cuda") * 0.03)
    bias = torch.zeros(c, device="cuda")
    out = HF.new_act(B, c, 2 * h, 2 * h, dt, "cuda")
    st = torch.zeros(HF.STAT_R, 2, c, dtype=torch.float64, device="cuda")
    wp = HF.PACKS.get(w, HF.L.PACK_CONVT_FWD, dt)
    fn = lambda: HF.conv_forward_raw(x, w, bias, 2, 1, transposed=True, out=out, stats=st, packed=wp)
    for _ in range(3):
        fn()
    HF.L.load().saunet_launch_log()
    fn()
    log = HF.L.load().saunet_launch_log().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * B * h * h * 16 * c * c
    print("%-22s %7.1f us  %7.1f TFLOP/s  %.3f of 2500  [%s]" % (name, us, fl / us / 1e6, fl / us / 1e6 / 2500, log))
