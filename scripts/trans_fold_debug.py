import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import saunet_amd as S
HF = S.functional
torch.manual_seed(1)
n, c, h = 2, 128, 128
dt = torch.bfloat16
trans = S.modules._Transition(c, c // 2).cuda().train()
with torch.no_grad():
    trans.norm.weight.uniform_(0.5, 1.5); trans.norm.bias.uniform_(4.0, 6.0)
buf0 = torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
cot = None
res = {}
for fold in (True, False):
    HF.begin_step()
    buf = buf0.clone().requires_grad_(True)
    st = HF.bn_stats(buf)
    if fold:
        st._saunet_fused_block = (buf.data_ptr(), float(trans.norm.eps))
    y = trans(buf, st)
    if cot is None:
        cot = torch.randn(y.shape, device="cuda").to(dt)
    (y.float() * cot.float()).sum().backward()
    g = buf.grad.float()
    if fold:
        ab = HF._PENDING_AB.pop(buf.data_ptr())
        count = n * h * h
        A = ab[:, 0].sum(0) / count; B = ab[:, 1].sum(0) / count
        x = buf0.double()
        mean = x.mean((0, 2, 3)); var = x.var((0, 2, 3), unbiased=False); inv = 1 / torch.sqrt(var + 1e-5)
        xhat = (x - mean.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1)
        g = (g.double() - A.view(1, -1, 1, 1) - B.view(1, -1, 1, 1) * xhat).float()
        print("A range", float(A.abs().max()), "B range", float(B.abs().max()), "|scale*g| rms", float(buf.grad.float().pow(2).mean().sqrt()))
    res[fold] = g
d = (res[True] - res[False])
print("rel L2 folded-corrected vs unfolded:", float(d.norm() / res[False].norm()))
print("per-channel pixel sums: unfolded max |sum| %.4f   folded max |sum| %.4f   rms value %.4f" % (float(res[False].sum((0, 2, 3)).abs().max()), float(res[True].sum((0, 2, 3)).abs().max()), float(res[False].pow(2).mean().sqrt())))
