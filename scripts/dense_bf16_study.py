"""Where does the bf16 dense-block gradient error come from?  HIP bf16 block vs (A) exact float64 and (B) float64 arithmetic with
bf16 rounding at the HIP path's storage / operand points (exact gradient accumulation).  Prints per-layer relative L2 errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import saunet_amd as S


class QFwd(torch.autograd.Function):            # operand rounding: forward only
    @staticmethod
    def forward(ctx, x): return x.to(torch.bfloat16).to(x.dtype)
    @staticmethod
    def backward(ctx, g): return g


class QBoth(torch.autograd.Function):           # stored tensor: value and its gradient live in bf16
    @staticmethod
    def forward(ctx, x): return x.to(torch.bfloat16).to(x.dtype)
    @staticmethod
    def backward(ctx, g): return g.to(torch.bfloat16).to(g.dtype)


def ref_block(block, x, emulate):
    d = torch.float64
    prm = {k: v.detach().to(d).requires_grad_(True) for k, v in block.named_parameters()}
    xr = x.detach().to(d).requires_grad_(True)
    qf = QFwd.apply if emulate else (lambda t: t)
    qb = QBoth.apply if emulate else (lambda t: t)
    feats = [xr]
    for name, layer in block.items():
        cat = torch.cat(feats, 1)
        a = qf(F.relu(F.batch_norm(cat, None, None, prm[name + ".norm1.weight"], prm[name + ".norm1.bias"], True, 0.0, layer.norm1.eps)))
        z1 = qb(F.conv2d(a, qf(prm[name + ".conv1.weight"])))
        b = qf(F.relu(F.batch_norm(z1, None, None, prm[name + ".norm2.weight"], prm[name + ".norm2.bias"], True, 0.0, layer.norm2.eps)))
        feats.append(qb(F.conv2d(b, qf(prm[name + ".conv2.weight"]), padding=1)))
    return torch.cat(feats, 1), xr, prm


def rel_l2(a, b):
    b = b.double(); a = a.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosv(a, b):
    a = a.detach().double().reshape(-1); b = b.double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


for layers, cin, shape in [(6, 64, (4, 64, 64)), (12, 128, (8, 32, 32)), (12, 256, (8, 32, 32))]:
    torch.manual_seed(layers * 100 + cin)
    n, h, w = shape
    block = S.modules._DenseBlock(layers, cin).cuda().train()
    with torch.no_grad():
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    x = torch.randn(n, cin, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = block(x)
    cot = torch.randn(y.shape, device="cuda").to(torch.bfloat16)
    (y.float() * cot.float()).sum().backward()
    res = {}
    for emu in (False, True):
        ry, xr, prm = ref_block(block, x, emu)
        (ry * cot.double()).sum().backward()
        res[emu] = (ry.detach(), xr.grad.detach(), {k: v.grad.detach() for k, v in prm.items()})
    A, B = res[False], res[True]
    print("=== layers %d cin %d shape %s" % (layers, cin, shape))
    print("  y   : hip-A %.4f  hip-B %.4f  B-A %.4f" % (rel_l2(y, A[0]), rel_l2(y, B[0]), rel_l2(B[0], A[0])))
    print("  dx  : hip-A %.4f  hip-B %.4f  B-A %.4f   cos hip-A %.4f" % (rel_l2(x.grad, A[1]), rel_l2(x.grad, B[1]), rel_l2(B[1], A[1]), cosv(x.grad, A[1])))
    for name in block.keys():
        row = []
        for suffix in ("norm1.weight", "norm1.bias", "conv1.weight", "norm2.weight", "norm2.bias", "conv2.weight"):
            k = name + "." + suffix
            g = dict(block.named_parameters())[k].grad
            row.append("%s %.3f/%.3f/%.3f" % (suffix.replace("weight", "w").replace("bias", "b"), rel_l2(g, A[2][k]), rel_l2(g, B[2][k]), rel_l2(B[2][k], A[2][k])))
        print("  %-12s (hip-A/hip-B/B-A) %s" % (name, "  ".join(row)))
