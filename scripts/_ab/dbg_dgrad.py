import sys, os
sys.path.insert(0, os.getcwd())
import torch
import saunet_amd as S
HF = S.functional
torch.manual_seed(1)
n, cin, h, k = 2, 128, 32, 128
dt = torch.float32
buf = torch.randn(n, 160, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
dbuf = torch.randn(n, 160, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
g = torch.randn(n, k, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(k, cin, 1, 1, device="cuda") * 0.05)
p = HF.BNParams(cin, "cuda")
p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
for acc in (False, True):
    sums = torch.zeros(HF.STAT_R, 2, cin, dtype=torch.float64, device="cuda")
    d0 = dbuf.clone()
    HF.conv_dgrad_raw(g, w, (n, cin, h, h), 1, 0, out=d0[:, :cin], bn_epi=(buf[:, :cin], p, True, sums, acc))
    torch.cuda.synchronize()
    # reference
    gg = torch.nn.functional.conv_transpose2d(g.double(), w.detach().double())      # dx = W^T g
    x = buf[:, :cin].double()
    mask = (x * p.scale.double().view(1, -1, 1, 1) + p.shift.double().view(1, -1, 1, 1)) > 0
    gm = gg * mask
    want = dbuf[:, :cin].double() + p.scale.double().view(1, -1, 1, 1) * gm if acc else gm
    xh = (x - p.mean.double().view(1, -1, 1, 1)) * p.invstd.double().view(1, -1, 1, 1)
    s = sums.sum(0)
    print("accumulate", acc, "out err %.3e" % float((d0[:, :cin].double() - want).abs().max() / want.abs().max()),
          "sum g err %.3e" % float((s[0] - gm.sum((0, 2, 3))).abs().max() / gm.sum((0, 2, 3)).abs().max()),
          "sum g*xhat err %.3e" % float((s[1] - (gm * xh).sum((0, 2, 3))).abs().max() / (gm * xh).sum((0, 2, 3)).abs().max()))
