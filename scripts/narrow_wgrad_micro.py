"""Weight gradients of the few-output 1x1 layers (c3/c4/c5/phi/cw/fuse/final) at the bench geometry: us per call, GB/s of algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
B = 32
for cin, cout, h, dt in ((2, 1, 256, torch.float32), (8, 1, 256, torch.bfloat16), (32, 4, 256, torch.bfloat16), (16, 1, 128, torch.bfloat16), (32, 1, 64, torch.bfloat16),
                         (64, 1, 32, torch.bfloat16), (256, 1, 32, torch.bfloat16), (128, 1, 16, torch.bfloat16), (1024, 1, 16, torch.bfloat16)):
    x = torch.randn(B, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.zeros(cout, cin, 1, 1, device="cuda"))
    fn = lambda: (HF.GRADS.reset(), HF.conv_wgrad_raw(x, dy, w, 1, 0))
    for _ in range(3):
        fn()
    HF.L.load().saunet_launch_log(); fn(); log = HF.L.load().saunet_launch_log().decode()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    by = B * h * h * (cin + cout) * (2 if dt == torch.bfloat16 else 4)
    print("%4d -> %d @%3d %-8s %7.1f us  %7.0f GB/s  [%s]" % (cin, cout, h, str(dt).split(".")[1], us, by / us / 1e3, log))
