"""Fused GatedSpatialConv2d forward + backward at the step's geometry (B = 32, 256 x 256, C = 32 / 16 / 8); run under rocprofv3 --kernel-trace --stats
for the per-kernel split.  usage: gate_micro.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for c in (32, 16, 8):
    m = S.GatedSpatialConv2d(c, c).cuda().train()
    feat = torch.randn(32, c, 256, 256, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gate = torch.randn(32, 1, 256, 256, device="cuda").to(torch.bfloat16).requires_grad_(True)
    wy = torch.randn(32, c, 256, 256, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wa = torch.randn(32, 1, 256, 256, device="cuda").to(torch.bfloat16)
    def step():
        y, a = m(feat, gate)
        torch.autograd.backward([y, a], [wy, wa])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record(); torch.cuda.synchronize()
    print("C=%d  fwd+bwd %.1f us" % (c, e0.elapsed_time(e1) * 1e3 / iters))
