"""Run ONE conv geometry a few times (for rocprofv3 PMC passes): python scripts/one_kernel.py conv2|conv1|res1|dec2..dec5|center|mrfup5|mrfup3|conv2dgrad|conv1dgrad [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
HF = S.functional
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = torch.bfloat16
n = 32
if which in ("conv1dgrad3", "conv1dgrad4", "conv2dgrad3", "conv2dgrad4"):     # the same on the low-resolution blocks (32 x 32 / 16 x 16 maps)
    h = 32 if which.endswith("3") else 16
    cin, ctot = (640, 1024) if h == 32 else (768, 1024)
    buf = torch.randn(n, ctot, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    dbuf = torch.randn(n, ctot, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    z1 = torch.randn(n, 128, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    c = 128 if which.startswith("conv2") else cin
    p = HF.BNParams(c, "cuda"); p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    if which.startswith("conv2"):
        w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.05)
        for _ in range(reps):
            HF.conv_dgrad_raw(dbuf[:, cin:cin + 32], w, z1.shape, 1, 1, bn_epi=(z1, p, True, HF.new_stats(128, "cuda")))
    else:
        w = torch.nn.Parameter(torch.randn(128, cin, 1, 1, device="cuda") * 0.05)
        for _ in range(reps):
            HF.conv_dgrad_raw(z1, w, (n, cin, h, h), 1, 0, out=dbuf[:, :cin], bn_epi=(buf[:, :cin], p, True, HF.new_stats(cin, "cuda"), True))
    torch.cuda.synchronize()
    sys.exit(0)
if which in ("conv2dgrad", "conv1dgrad"):        # DenseNet data gradients with their BatchNorm-backward reduction epilogues, block-1 geometry
    dbuf = torch.randn(n, 256, 128, 128, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    z1 = torch.randn(n, 128, 128, 128, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    p = HF.BNParams(128 if which == "conv2dgrad" else 192, "cuda")
    p.buf[0].uniform_(0.5, 1.5); p.buf[1].normal_(0, 0.3); p.buf[2].normal_(0, 0.3); p.buf[3].uniform_(0.5, 1.5)
    if which == "conv2dgrad":
        w = torch.nn.Parameter(torch.randn(32, 128, 3, 3, device="cuda") * 0.05)
        for _ in range(reps):
            HF.conv_dgrad_raw(dbuf[:, 64:96], w, z1.shape, 1, 1, bn_epi=(z1, p, True, HF.new_stats(128, "cuda")))
    else:
        w = torch.nn.Parameter(torch.randn(128, 192, 1, 1, device="cuda") * 0.05)
        xin = torch.randn(n, 256, 128, 128, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        for _ in range(reps):
            HF.conv_dgrad_raw(z1, w, (n, 192, 128, 128), 1, 0, out=dbuf[:, :192], bn_epi=(xin[:, :192], p, True, HF.new_stats(192, "cuda"), True))
    torch.cuda.synchronize()
    sys.exit(0)
if which == "conv2":
    cin, h, cout, k, pro = 128, 128, 32, 3, True
elif which == "conv1":
    cin, h, cout, k, pro = 160, 128, 128, 1, True
elif which == "res1":
    cin, h, cout, k, pro = 64, 256, 64, 3, False
elif which == "dec3":
    cin, h, cout, k, pro = 512, 64, 128, 3, False
elif which == "dec2":
    cin, h, cout, k, pro = 256, 128, 64, 3, False
elif which == "dec4":
    cin, h, cout, k, pro = 1024, 32, 256, 3, False
elif which == "dec5":
    cin, h, cout, k, pro = 1536, 16, 512, 3, False
elif which == "center":                       # conv3x3_bn_relu(1024, 512) on the 8x8 map (im2col + pointwise path)
    cin, h, cout, k, pro = 1024, 8, 512, 3, False
elif which == "mrfup5":                       # dec5.mrf.up: ConvTranspose2d(512, 512, 4, 2, 1) 8x8 -> 16x16
    cin, h, cout, k, pro = 512, 8, 512, 4, False
elif which == "mrfup3":                       # dec3.mrf.up: ConvTranspose2d(128, 128, 4, 2, 1) 32x32 -> 64x64
    cin, h, cout, k, pro = 128, 32, 128, 4, False
x = torch.randn(n, cin, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
if k == 4:                                    # transposed convolution: weight [Cin, Cout, 4, 4], output 2h x 2h
    w = torch.nn.Parameter(torch.randn(cin, cout, 4, 4, device="cuda") * 0.03)
    out = HF.new_act(n, cout, 2 * h, 2 * h, dt, "cuda")
    st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    for _ in range(reps):
        HF.conv_forward_raw(x, w, None, 2, 1, transposed=True, out=out, stats=st)
    torch.cuda.synchronize()
    sys.exit(0)
w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.03)
sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.1
out = HF.new_act(n, cout, h, h, dt, "cuda")
st = torch.zeros(HF.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
for _ in range(reps):
    HF.conv_forward_raw(x, w, None, 1, k // 2, pro=(sc, sh, True) if pro else None, out=out, stats=st)
torch.cuda.synchronize()
