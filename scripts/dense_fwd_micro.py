"""Forward of DenseNet blocks 3 / 4 at the bench geometry (B=32, 256x256 input): per-layer launches vs the persistent whole-block kernel.
python scripts/dense_fwd_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
S.set_compute_dtype(torch.bfloat16)


def bench(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, layers, cin, shape in (("block3", 24, 256, (32, 32, 32)), ("block4", 16, 512, (32, 16, 16)), ("block2", 12, 128, (32, 64, 64))):
    n, h, w = shape
    block = S.modules._DenseBlock(layers, cin).cuda().train()
    x = torch.randn(n, cin, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for mode in (False, True):
        HF.DENSE_PERSIST = mode
        HF.DENSE_PERSIST_MAXPIX = 1 << 30

        def run():
            with torch.no_grad():
                block(x)
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            run(); run()
        torch.cuda.current_stream().wait_stream(s_)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        us = bench(g.replay)
        print("%s  persistent=%d  %8.1f us forward in a graph (%d layers, %.1f us/layer)" % (name, mode, us, layers, us / layers), flush=True)
