"""DenseNet conv1 dgrad + BN-backward epilogue micro-benchmark (the dominant kernel): python scripts/dgrad_micro.py [cin] [hw] [batch] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 192
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dt = torch.bfloat16
ctot = max(256, cin)
buf = torch.randn(n, ctot, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)     # concat buffer (x)
dbuf = torch.randn(n, ctot, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)    # gradient buffer (y)
g = torch.randn(n, 128, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(128, cin, 1, 1, device="cuda") * 0.05)
p = HF.BNParams(cin, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
def run():
    st = HF.new_stats(cin, "cuda")
    HF.conv_dgrad_raw(g, w, (n, cin, hw, hw), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, st, True))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
P = n * hw * hw
byts = (128 + 3 * cin) * 2 * P
print("cin=%d P=%d  %.1f us  %.2f TB/s algorithmic  (%.0f MB)  %.1f TF/s" % (cin, P, ms * 1e3, byts / ms / 1e9, byts / 1e6, 2.0 * P * 128 * cin / ms / 1e9))
