import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_hip_dense as T
import saunet_amd as S
dtype = torch.float32
layers, cin, (n, h, w) = 3, 64, (2, 32, 32)
torch.manual_seed(layers * 100 + cin)
block = S.modules._DenseBlock(layers, cin).cuda().train()
with torch.no_grad():
    for m in block.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
x = torch.randn(n, cin, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = block(x)
cot = torch.randn(y.shape, device="cuda").to(dtype)
(y.float() * cot.float()).sum().backward()
ry, xr, prm = T.ref_block(block, x)
(ry * cot.double()).sum().backward()
print("y", T.rel(y, ry))
print("x.grad", T.rel(x.grad, xr.grad), "l2", T.rel_l2(x.grad, xr.grad))
d = (x.grad.double() - xr.grad).abs()
print(" per-channel max err of x.grad:", [round(float(v), 5) for v in d.amax((0, 2, 3))[:16]], "...")
print(" nan?", bool(torch.isnan(x.grad).any()))
for k, v in block.named_parameters():
    print(k, "%.2e" % T.rel(v.grad, prm[k].grad))
for name in ("denselayer3.norm1.bias", "denselayer3.norm1.weight", "denselayer2.conv2.weight"):
    g = dict(block.named_parameters())[name].grad.double().cpu().flatten(); r = prm[name].grad.cpu().flatten()
    e = (g - r).abs()
    idx = torch.argsort(e, descending=True)[:8]
    print(name, "scale %.3e" % float(r.abs().max()), "worst:", [(int(i), "%.3e" % float(e[i]), "%.4f" % float(g[i]), "%.4f" % float(r[i])) for i in idx])
yy = y.detach().double().cpu(); rr = ry.detach().cpu()
for c in (113, 112, 100):
    v = rr[:, c]
    print("channel", c, "mean %.4e var %.4e  max|y-ref| %.3e  rel-to-std %.3e" % (float(v.mean()), float(v.var()), float((yy[:, c] - v).abs().max()), float((yy[:, c] - v).abs().max() / v.std())))
l3 = block.denselayer3
print("norm1 gamma/beta ch113:", float(l3.norm1.weight[113]), float(l3.norm1.bias[113]))
# fraction of pixels whose layer-3 norm1 pre-activation is within 1e-5 of zero for channel 113
m = rr[:, 113].mean(); s = (rr[:, 113].var(unbiased=False) + l3.norm1.eps) ** 0.5
pre = (rr[:, 113] - m) / s * float(l3.norm1.weight[113]) + float(l3.norm1.bias[113])
print("pre-activation ch113: min %.3e max %.3e  count |pre|<1e-4: %d of %d" % (float(pre.min()), float(pre.max()), int((pre.abs() < 1e-4).sum()), pre.numel()))
pre_h = (yy[:, 113] - yy[:, 113].mean()) / (yy[:, 113].var(unbiased=False) + l3.norm1.eps) ** 0.5 * float(l3.norm1.weight[113]) + float(l3.norm1.bias[113])
print("mask flips ch113 (HIP forward values vs reference):", int(((pre > 0) != (pre_h > 0)).sum()))
