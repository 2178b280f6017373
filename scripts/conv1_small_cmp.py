import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import saunet_amd as S
HF = S.functional
out = sys.argv[1]
n, h, cin, c_lo = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (8, 16, 544, 512)
dt = torch.bfloat16
g = torch.Generator().manual_seed(3)
buf = torch.randn(n, 1024, h, h, generator=g).cuda().to(dt).contiguous(memory_format=torch.channels_last)
x = buf[:, :cin]
weight = torch.nn.Parameter((torch.randn(128, cin, 1, 1, generator=g) * 0.05).cuda())
gamma = torch.empty(cin).uniform_(0.5, 1.5, generator=g).cuda(); beta = torch.empty(cin).uniform_(-0.3, 0.3, generator=g).cuda()
count = n * h * h
HF.STATS.reset(); HF.GRADS.reset()
stats = HF.bn_stats(x)
HF.collapse_stats(stats); stats[1:] = 0        # one replica: the input statistics are bit-identical in both runs
xh = torch.zeros(5, 1024, dtype=torch.float32, device="cuda")
HF.L.call("saunet_bn_xhat", c_lo, stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), stats.shape[0], stats.stride(0), float(count), 1e-5, xh.data_ptr(), xh.stride(0), HF.L.stream())
params = HF.BNParams(cin, "cuda")
rm, rv = torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")
st_out = HF.new_stats(128, "cuda")
HF.L.load().saunet_launch_log()
y = HF.conv_forward_bnpro(x, weight, 1, 0, stats, count, c_lo, xh, gamma, beta, rm, rv, 0.1, 1e-5, params.buf, stats=st_out)
print(HF.L.load().saunet_launch_log().decode())
sums = HF.collapse_stats(st_out).clone()
torch.cuda.synchronize()
torch.save({"y": y.float().cpu(), "sums": sums.cpu(), "params": params.buf.cpu(), "xh": xh.cpu()}, out)
