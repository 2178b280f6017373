"""How much of the shape stream hides behind the encoder / decoder when they run on two HIP streams?  Two captured graphs replayed alone and
together: A = dense block (latency-bound chain) forward + backward, or a decoder-like MFMA-bound 3x3 conv chain; B = BasicBlock(64) forward +
backward at full resolution (HBM-bound).   python scripts/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
dt = torch.bfloat16
n = 32
torch.manual_seed(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def act(c, h, grad=False):
    t = torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    return t.requires_grad_(True) if grad else t


def capture(fn, stream):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# A1: dense block 3 forward + backward
block = S.modules._DenseBlock(24, 256).cuda().train()
xb = act(256, 32, True); cot = act(256 + 32 * 24, 32)
def dense():
    block.zero_grad(set_to_none=True); xb.grad = None
    block(xb).backward(cot)
# A2: decoder-like chain: dec3 + dec4 c3x3rb forward, dgrad, wgrad
xd3, wd3 = act(512, 64), torch.nn.Parameter(torch.randn(128, 512, 3, 3, device="cuda") * 0.02)
xd4, wd4 = act(1024, 32), torch.nn.Parameter(torch.randn(256, 1024, 3, 3, device="cuda") * 0.02)
dy3, dy4 = act(128, 64), act(256, 32)
def decoder():
    for x_, w_, dy_ in ((xd3, wd3, dy3), (xd4, wd4, dy4)):
        HF.conv_forward_raw(x_, w_, None, 1, 1)
        HF.conv_dgrad_raw(dy_, w_, x_.shape, 1, 1)
        HF.conv_wgrad_raw(x_, dy_, w_, 1, 1)
# B: BasicBlock(64) at full resolution, forward + backward
res = S.BasicBlock(64, 64).cuda().train()
xr = act(64, 256, True); cotr = act(64, 256)
def shape():
    res.zero_grad(set_to_none=True); xr.grad = None
    res(xr).backward(cotr)

# separate scratch arenas would be needed for truly concurrent eager use; the graphs below each own their captured allocations
gA1 = capture(lambda: (HF.STATS.reset(), HF.GRADS.reset(), dense()), s1)
gA2 = capture(lambda: (HF.STATS.reset(), HF.GRADS.reset(), decoder()), s1)
gB = capture(lambda: (HF.STATS.reset(), HF.GRADS.reset(), shape()), s2)


def alone(g, s):
    def f():
        with torch.cuda.stream(s):
            g.replay()
        torch.cuda.current_stream().wait_stream(s)
    return f


def both(ga):
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ga.replay()
        with torch.cuda.stream(s2):
            gB.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f


tA1, tA2, tB = timeit(alone(gA1, s1)), timeit(alone(gA2, s1)), timeit(alone(gB, s2))
t1, t2 = timeit(both(gA1)), timeit(both(gA2))
print("dense block 3 fwd+bwd alone %.2f ms | decoder 3x3 chain alone %.2f ms | BasicBlock(64) @256^2 fwd+bwd alone %.2f ms" % (tA1, tA2, tB))
print("dense  || shape: %.2f ms (sum %.2f, max %.2f): %.0f %% of the shorter one hidden" % (t1, tA1 + tB, max(tA1, tB), 100 * (tA1 + tB - t1) / min(tA1, tB)))
print("decoder|| shape: %.2f ms (sum %.2f, max %.2f): %.0f %% of the shorter one hidden" % (t2, tA2 + tB, max(tA2, tB), 100 * (tA2 + tB - t2) / min(tA2, tB)))
