"""GPU parity of the fused GatedSpatialConv2d kernels (csrc/gate.hip) against a float64 restatement of the module
(/root/reference/models/GSConv.py:16-57) evaluated on the SAME bf16-rounded inputs and float32 parameters.  The fused
path keeps every intermediate in float32 accumulators (the per-pixel products run on the matrix cores with the float32 weights as
bf16 hi + lo operand pairs and the bf16 activations exactly as stored: ~2^-17 relative), so outputs differ from the float64 result
only by their final bf16 rounding (2^-9 relative); the second backward product (W1^T dh) and the cross-pixel sums use single bf16
matrix-core operands (input / weight gradients: ~1e-2 of the tensor's scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def reference(m, feat, gate):
    g = m._gate_conv
    d = torch.float64
    p = lambda t: t.detach().to(d).requires_grad_(True)
    prm = dict(g0w=p(g[0].weight), g0b=p(g[0].bias), w1=p(g[1].weight), b1=p(g[1].bias), w2=p(g[3].weight), b2=p(g[3].bias),
               g1w=p(g[4].weight), g1b=p(g[4].bias), wm=p(m.weight))
    f, ga = p(feat), p(gate)
    a = F.batch_norm(torch.cat([f, ga], 1), None, None, prm["g0w"], prm["g0b"], True, 0.0, g[0].eps)
    h = F.relu(F.conv2d(a, prm["w1"], prm["b1"]))
    z = F.conv2d(h, prm["w2"], prm["b2"])
    alpha = torch.sigmoid(F.batch_norm(z, None, None, prm["g1w"], prm["g1b"], True, 0.0, g[4].eps))
    y = F.conv2d(f * (alpha + 1), prm["wm"])
    return y, alpha, f, ga, prm


def rel(a, b):
    b = b.to(torch.float64)
    return float((a.to(torch.float64) - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("c,shape", [(32, (2, 32, 32)), (16, (3, 20, 12)), (8, (1, 16, 16)), (32, (1, 8, 24))])
def test_fused_gate_matches_float64(c, shape):
    import saunet_amd as S
    torch.manual_seed(100 + c + shape[1])
    n, h, w = shape
    m = S.GatedSpatialConv2d(c, c).cuda().train()
    with torch.no_grad():
        for bn in (m._gate_conv[0], m._gate_conv[4]):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        m._gate_conv[3].weight.mul_(3.0)
    feat = (torch.randn(n, c, h, w, device="cuda") * 1.5 + 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gate = torch.randn(n, 1, h, w, device="cuda").to(torch.bfloat16).requires_grad_(True)
    assert S.functional.gated_conv_fusable(feat, gate, m)
    y, alpha = m(feat, gate)
    wy = torch.randn_like(y, dtype=torch.float32).to(torch.bfloat16)
    wa = torch.randn_like(alpha, dtype=torch.float32).to(torch.bfloat16)
    ((y * wy).sum() + (alpha * wa).sum()).backward()

    ry, ralpha, rf, rg, prm = reference(m, feat, gate)
    ((ry * wy.double()).sum() + (ralpha * wa.double()).sum()).backward()
    assert rel(y, ry) < 1e-2 and rel(alpha, ralpha) < 1e-2
    assert rel(feat.grad, rf.grad) < 2e-2, rel(feat.grad, rf.grad)
    assert rel(gate.grad, rg.grad) < 2e-2, rel(gate.grad, rg.grad)
    g = m._gate_conv
    got = dict(g0w=g[0].weight.grad, g0b=g[0].bias.grad, w1=g[1].weight.grad, b1=g[1].bias.grad, w2=g[3].weight.grad, b2=g[3].bias.grad,
               g1w=g[4].weight.grad, g1b=g[4].bias.grad, wm=m.weight.grad)
    for k, v in got.items():
        if k == "b2":   # a bias in front of a training-mode batch norm: exactly zero gradient, only rounding noise is left
            assert float(v.abs().max()) < 1e-2 * float(prm["w2"].grad.abs().max())
            continue
        assert rel(v, prm[k].grad) < 2e-2, (k, rel(v, prm[k].grad))
    # running statistics follow the batch statistics of cat and z
    cat = torch.cat([feat.detach().double(), gate.detach().double()], 1)
    assert torch.allclose(g[0].running_mean.double(), 0.1 * cat.mean((0, 2, 3)), atol=1e-3)


def test_fused_gate_eval_forward_uses_running_stats():
    import saunet_amd as S
    torch.manual_seed(7)
    m = S.GatedSpatialConv2d(16, 16).cuda().eval()
    with torch.no_grad():
        m._gate_conv[0].running_mean.uniform_(-0.2, 0.2); m._gate_conv[0].running_var.uniform_(0.5, 1.5)
        m._gate_conv[4].running_mean.fill_(0.1); m._gate_conv[4].running_var.fill_(0.7)
        feat = torch.randn(2, 16, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gate = torch.randn(2, 1, 16, 16, device="cuda").to(torch.bfloat16)
        assert S.functional.gated_conv_fusable(feat, gate, m)
        y, alpha = m(feat, gate)
        g = m._gate_conv
        f, ga = feat.double(), gate.double()
        a = F.batch_norm(torch.cat([f, ga], 1), g[0].running_mean.double(), g[0].running_var.double(), g[0].weight.double(), g[0].bias.double(), False, 0.0, g[0].eps)
        z = F.conv2d(F.relu(F.conv2d(a, g[1].weight.double(), g[1].bias.double())), g[3].weight.double(), g[3].bias.double())
        ra = torch.sigmoid(F.batch_norm(z, g[4].running_mean.double(), g[4].running_var.double(), g[4].weight.double(), g[4].bias.double(), False, 0.0, g[4].eps))
        ry = F.conv2d(f * (ra + 1), m.weight.double())
    assert rel(y, ry) < 1e-2 and rel(alpha, ra) < 1e-2


@pytest.mark.parametrize("out_dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("shape", [(2, 32, 40), (1, 24, 24)])
def test_fused_expand_matches_float64(out_dtype, tol, shape):
    """SAUNet.expand: Conv2d(1, 32, 1x1) -> BatchNorm -> ReLU on a one-channel float32 map (csrc/expand.hip)."""
    import saunet_amd as S
    torch.manual_seed(11)
    n, (h, w) = shape[0], shape[1:]
    m = S.ConvBNReLU(1, 32, kernel_size=1, padding=0).cuda().train()
    with torch.no_grad():
        m[1].weight.uniform_(0.5, 1.5); m[1].bias.uniform_(-0.5, 0.5); m[0].weight.normal_(0, 1.0); m[0].bias.normal_(0, 0.3)
    a = torch.rand(n, 1, h, w, device="cuda").requires_grad_(True)
    assert S.functional.expand_fusable(a, m[0], m[1])
    y = m(a, out_dtype=out_dtype)
    assert y.dtype == out_dtype
    cot = torch.randn(y.shape, device="cuda").to(out_dtype)
    (y.float() * cot.float()).sum().backward()
    d = torch.float64
    p = lambda t: t.detach().to(d).requires_grad_(True)
    ra, rw, rb, rg, rbeta = p(a), p(m[0].weight), p(m[0].bias), p(m[1].weight), p(m[1].bias)
    ry = F.relu(F.batch_norm(F.conv2d(ra, rw, rb), None, None, rg, rbeta, True, 0.0, m[1].eps))
    (ry * cot.double()).sum().backward()
    assert rel(y, ry) < tol
    assert rel(a.grad, ra.grad) < 10 * tol, rel(a.grad, ra.grad)
    assert rel(m[0].weight.grad, rw.grad) < 10 * tol
    assert rel(m[1].weight.grad, rg.grad) < 10 * tol and rel(m[1].bias.grad, rbeta.grad) < 10 * tol
    assert float(m[0].bias.grad.abs().max()) == 0.0
    conv_out = F.conv2d(a.detach().double(), m[0].weight.double(), m[0].bias.double())
    assert torch.allclose(m[1].running_mean.double(), 0.1 * conv_out.mean((0, 2, 3)), atol=1e-5)
    assert torch.allclose(m[1].running_var.double(), 0.9 + 0.1 * conv_out.var((0, 2, 3), unbiased=True), atol=1e-4)
