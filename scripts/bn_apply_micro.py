"""Streaming BatchNorm kernels against the box's own elementwise ceiling: bn_backward apply (in place, with pre-computed sums) and reduce, affine_act,
next to ATen's triad / in-place add over the same bytes.  python scripts/bn_apply_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
dt = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
K = 8          # rotate over K operand sets: 3 x 134 MB x 8 = 3.2 GB at the first geometry, far beyond the 256 MB memory-side cache
for (n, c, h) in ((32, 128, 128), (32, 128, 64), (32, 32, 256), (32, 64, 128), (32, 512, 16)):
    G = [torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last) for _ in range(K)]
    X = [torch.randn(n, c, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last) for _ in range(K)]
    O = [torch.empty_like(G[0]) for _ in range(K)]
    it = [0]
    def nxt():
        it[0] = (it[0] + 1) % K
        return G[it[0]], X[it[0]], O[it[0]]
    p = HF.BNParams(c, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    st = HF.new_stats(c, "cuda"); st.normal_()
    P = n * h * h; by = P * c * 2
    g, x, o = G[0], X[0], O[0]
    def f1():
        g, x, o = nxt(); HF.bn_backward(g, x, p, True, P, True, dx=g, presums=st)
    def f2():
        g, x, o = nxt(); HF.bn_backward(g, x, p, True, P, True, dx=o, presums=st)
    def f3():
        g, x, o = nxt(); HF.bn_backward(g, x, p, True, P, True, dx=o)
    def f4():
        g, x, o = nxt(); HF.affine_act(x, p.scale, p.shift, True, out=o)
    def f5():
        g, x, o = nxt(); torch.add(g, x, out=o)
    def f6():
        g, x, o = nxt(); g.add_(x)
    def f7():
        g, x, o = nxt(); o.copy_(x)
    from saunet_amd import lib as L
    def f0():
        g, x, o = nxt()
        L.call("saunet_bn_backward_reduce", L.dtype_code(x), g.data_ptr(), HF.ld_of(g), x.data_ptr(), HF.ld_of(x), None, 0, p.scale.data_ptr(), p.shift.data_ptr(),
               p.mean.data_ptr(), p.invstd.data_ptr(), 1, st.data_ptr(), st.shape[0], st.stride(0), P, c, L.stream())
    ms = t(f0)
    print("%-22s reduce          %7.1f us  %.2f TB/s" % ((n, c, h), ms * 1e3, 2 * by / ms / 1e9))
    ms = t(f1)
    print("%-22s apply in place  %7.1f us  %.2f TB/s" % ((n, c, h), ms * 1e3, 3 * by / ms / 1e9))
    ms = t(f2)
    print("%-22s apply           %7.1f us  %.2f TB/s" % ("", ms * 1e3, 3 * by / ms / 1e9))
    ms = t(f3)
    print("%-22s reduce + apply  %7.1f us  %.2f TB/s" % ("", ms * 1e3, 5 * by / ms / 1e9))
    ms = t(f4)
    print("%-22s affine_act      %7.1f us  %.2f TB/s" % ("", ms * 1e3, 2 * by / ms / 1e9))
    ms = t(f5); print("%-22s ATen triad      %7.1f us  %.2f TB/s" % ("", ms * 1e3, 3 * by / ms / 1e9))
    ms = t(f6); print("%-22s ATen in place   %7.1f us  %.2f TB/s" % ("", ms * 1e3, 3 * by / ms / 1e9))
    ms = t(f7); print("%-22s ATen copy       %7.1f us  %.2f TB/s" % ("", ms * 1e3, 2 * by / ms / 1e9))
