"""Would running the DenseNet conv1 data gradient and weight gradient of a layer CONCURRENTLY pay?  Times both kernels back to back on one
stream and on two streams (eager launches, HIP events) at the four dense-block geometries of the bench step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
dt = torch.bfloat16
n = 32
def act(c, h): return torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream()
for blk, (h, cin, ctot) in enumerate(((128, 192, 256), (64, 384, 512), (32, 640, 1024), (16, 768, 1024)), 1):
    buf = act(ctot, h); dbuf = act(ctot, h); g = act(128, h)
    w = torch.nn.Parameter(torch.randn(128, cin, 1, 1, device="cuda") * 0.05)
    p = HF.BNParams(cin, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    sums = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda")
    dgrad = lambda: HF.conv_dgrad_raw(g, w, (n, cin, h, h), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, sums, True))
    wgrad = lambda: (HF.GRADS.reset(), HF._conv_wgrad_impl(buf[:, :cin], g, w, 1, 0, pro=(p.scale, p.shift, True)))
    def timed(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    def both_seq(): dgrad(); wgrad()
    def both_par():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        main.wait_stream(side)
    td, tw, ts, tp = timed(dgrad), timed(wgrad), timed(both_seq), timed(both_par)
    print("block%d Cin=%d @%d: dgrad %.1f us  wgrad %.1f us  sequential %.1f us  two streams %.1f us" % (blk, cin, h, td, tw, ts, tp), flush=True)
