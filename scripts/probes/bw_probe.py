import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (134, 268, 537):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x); z = torch.empty_like(x)
    ms = t(lambda: y.copy_(x)); print("copy   %4d MB  %.1f us  %.2f TB/s" % (mb, ms * 1e3, 2 * n * 2 / ms / 1e9))
    ms = t(lambda: torch.add(x, y, out=z)); print("triad  %4d MB  %.1f us  %.2f TB/s" % (mb, ms * 1e3, 3 * n * 2 / ms / 1e9))
    ms = t(lambda: x.add_(y)); print("inplace %4d MB  %.1f us  %.2f TB/s" % (mb, ms * 1e3, 3 * n * 2 / ms / 1e9))
    ms = t(lambda: x.sum()); print("read   %4d MB  %.1f us  %.2f TB/s" % (mb, ms * 1e3, n * 2 / ms / 1e9))
