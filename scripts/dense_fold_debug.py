import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import saunet_amd as S
from test_hip_dense import ref_block, rel_l2
HF = S.functional
layers, cin, shape = int(sys.argv[1]), int(sys.argv[2]), tuple(int(v) for v in sys.argv[3:6])
torch.manual_seed(31 + layers)
n, h, w = shape
dtype = torch.bfloat16
block = S.modules._DenseBlock(layers, cin).cuda().train()
ctot = cin + 32 * layers
trans = S.modules._Transition(ctot, ctot // 2).cuda().train()
with torch.no_grad():
    for m in list(block.modules()) + list(trans.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(4.0, 6.0)
x0 = torch.randn(n, cin, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
cot, grads = None, {}
captured = {}
_orig_g = HF.conv_wgrad_grouped
def cap(problems, ksize, pad, relu):
    if ksize == 3:
        captured[HF.DENSE_BWD_FUSED] = [dy.float().clone() for (_x, dy, _w, _p) in problems]
    return _orig_g(problems, ksize, pad, relu)
HF.conv_wgrad_grouped = cap
rec = {}
_orig_bf = HF._DenseBlock._backward_fused
def bf(ctx, buf, dbuf, xh, params, saved, grads, pending_ab=None):
    rec.update(dbuf=dbuf.float().clone(), ab=None if pending_ab is None else pending_ab.clone(), buf=buf.float().clone(), xh=xh.clone(), count=ctx.meta[3])
    return _orig_bf(ctx, buf, dbuf, xh, params, saved, grads, pending_ab)
HF._DenseBlock._backward_fused = staticmethod(bf)
names = [("block." + k, v) for k, v in block.named_parameters()] + [("trans." + k, v) for k, v in trans.named_parameters()]
for fused in (True, False):
    HF.DENSE_BWD_FUSED = fused
    HF.begin_step()
    for _, v in names: v.grad = None
    x = x0.clone().requires_grad_(True)
    buf, st = block(x, with_stats=True)
    y = trans(buf, st)
    if cot is None: cot = torch.randn(y.shape, device="cuda").to(dtype)
    (y.float() * cot.float()).sum().backward()
    grads[fused] = {"x": x.grad.float().clone(), **{k: v.grad.float().clone() for k, v in names}}
d = torch.float64
ry, xr, prm = ref_block(block, x0)
tp = {k: v.detach().to(d).requires_grad_(True) for k, v in trans.named_parameters()}
t = F.batch_norm(ry, None, None, tp["norm.weight"], tp["norm.bias"], True, 0.0, trans.norm.eps)
t = F.avg_pool2d(F.conv2d(F.relu(t), tp["conv.weight"]), 2)
(t * cot.double()).sum().backward()
ref = {"x": xr.grad, **{"block." + k: prm[k].grad for k in prm}, **{"trans." + k: tp[k].grad for k in tp}}
for k in grads[True]:
    print("%-36s fused-vs-unfused %.3e   fused-vs-f64 %.3e   unfused-vs-f64 %.3e   |ref| %.3e" % (k, rel_l2(grads[True][k], grads[False][k]), rel_l2(grads[True][k], ref[k]), rel_l2(grads[False][k], ref[k]), float(ref[k].norm())))

for i, (a, b) in enumerate(zip(captured[True], captured[False])):
    print("dz2 of problem %d: rel diff %.3e  pixel-sum max fused %.4f unfused %.4f  rms %.4f" % (i, float((a - b).norm() / b.norm()), float(a.sum((0, 2, 3)).abs().max()), float(b.sum((0, 2, 3)).abs().max()), float(b.pow(2).mean().sqrt())))

ab, xh, P = rec["ab"], rec["xh"], rec["count"]
lo = ctot - 32
A = (ab[:, 0, lo:].sum(0) / P); B = (ab[:, 1, lo:].sum(0) / P)
xhat = rec["buf"][:, lo:].double() * xh[0, lo:].double().view(1, -1, 1, 1) + xh[1, lo:].double().view(1, -1, 1, 1)
host = rec["dbuf"][:, lo:].double() - A.view(1, -1, 1, 1) - B.view(1, -1, 1, 1) * xhat
print("xhat channel means (should be ~0): max |mean| %.4e ; xhat rms %.3f" % (float(xhat.mean((0, 2, 3)).abs().max()), float(xhat.pow(2).mean().sqrt())))
print("host-corrected last chunk: pixel-sum max %.4f ; kernel dz2 pixel-sum max %.4f ; unfused %.4f" % (float(host.sum((0, 2, 3)).abs().max()), float(captured[True][0].sum((0, 2, 3)).abs().max()), float(captured[False][0].sum((0, 2, 3)).abs().max())))
print("uncorrected chunk pixel-sum max %.4f  P*A max %.4f" % (float(rec["dbuf"][:, lo:].sum((0, 2, 3)).abs().max()), float((A * P).abs().max())))
xd = x0.double()
k = captured[True][0].double()
dlt = (k - host)
ulp = 2.0 ** -9 * host.abs().clamp_min(1e-6)
print("kernel - host: mean %.3e  rms %.3e  | in ulps: mean %.3f rms %.3f  max %.2f" % (float(dlt.mean()), float(dlt.pow(2).mean().sqrt()), float((dlt / ulp).mean()), float((dlt / ulp).pow(2).mean().sqrt()), float((dlt / ulp).abs().max())))
hb = host.to(torch.bfloat16).double()
print("bf16(host) - host pixel-sum max %.4f ; kernel - bf16(host): nonzero frac %.4f, mean %.3e" % (float((hb - host).sum((0, 2, 3)).abs().max()), float((k != hb).double().mean()), float((k - hb).mean())))
ch = (k - hb).sum((0, 2, 3))
print("per-channel sum of kernel - bf16(host):", [round(float(v), 3) for v in ch[:8]], " B:", [round(float(v), 5) for v in B[:8]], " A:", [round(float(v), 5) for v in A[:8]])
