"""DenseNet conv2 (3x3, 128 -> 32) data gradient + norm2 BN-backward reduction epilogue (dense_dgrad3_kernel) at the step's four geometries:
python scripts/dgrad3_micro.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt, n = torch.bfloat16, 32
for hw in (128, 64, 32, 16):
    dbuf = torch.randn(n, 256, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    z1 = torch.randn(n, 128, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.05)
    p = HF.BNParams(128, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    def run():
        st = HF.new_stats(128, "cuda")
        return HF.conv_dgrad_raw(dbuf[:, 64:96], w, z1.shape, 1, 1, bn_epi=(z1, p, True, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    P = n * hw * hw
    byts = (32 + 128 + 128) * 2 * P
    print("dgrad3 @%dx%d P=%d  %.1f us  %.2f TB/s algorithmic  %.1f TF/s" % (hw, hw, P, ms * 1e3, byts / ms / 1e9, 2.0 * P * 288 * 128 / ms / 1e9))
