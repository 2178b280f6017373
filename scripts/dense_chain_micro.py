"""One DenseNet block (forward / forward + backward) as a captured hipGraph at the step's geometry: the latency-bound layer chain in isolation.
    python scripts/dense_chain_micro.py <block 1..4> [batch] [reps]
Prints us per replay for the forward graph and for forward + backward, and the per-layer figures (what VERDICT r4 item 1 asks about:
blocks 3 / 4 are 40 of the 58 layers and run 4-8 launches per layer with a 13-25 us floor each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
layers, c0, hw = {1: (6, 64, 128), 2: (12, 128, 64), 3: (24, 256, 32), 4: (16, 512, 16)}[blk]
torch.manual_seed(0)
block = S.modules._DenseBlock(layers, c0).cuda().train()
x = torch.randn(n, c0, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
cot = torch.randn(n, c0 + 32 * layers, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def fwd():
    HF.STATS.reset(); HF.GRADS.reset()
    with torch.no_grad():
        pass
    return block(x)


def fwd_bwd():
    HF.STATS.reset(); HF.GRADS.reset()
    block.zero_grad(set_to_none=True); x.grad = None
    y = block(x)
    y.backward(cot)
    return y


def timed(fn):
    g = S.graph.GraphedStep(fn, warmup=2, changes_params=False)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tf = timed(fwd)
tb = timed(fwd_bwd)
print("block %d (%d layers, %dx%d, B=%d, fused_bwd=%s): forward %.1f us (%.1f / layer)   forward+backward %.1f us   backward %.1f us (%.1f / layer)"
      % (blk, layers, hw, hw, n, HF.DENSE_BWD_FUSED, tf, tf / layers, tb, tb - tf, (tb - tf) / layers))
