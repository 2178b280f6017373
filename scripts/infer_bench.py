"""Inference throughput of the eval-mode forward (SegmentationModule inference branch: softmax + argmax-ready scores),
B x 256 x 256 bf16, captured in a hipGraph: python scripts/infer_bench.py [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
from saunet_amd import data

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = S.SAUNet(num_classes=4).cuda().eval()
img, seg, edge = data.synthetic_batch(b, size, size, seed=1)
x = img.cuda()
with torch.no_grad():
    for _ in range(3):
        logits, e = net(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        logits, e = net(x)
        probs = torch.softmax(logits.float(), 1)
    g.replay(); torch.cuda.synchronize()
    t = time.time(); n = 30
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    ms = (time.time() - t) / n * 1e3
print("inference (eval BN, hipGraph): B=%d %dx%d  %.2f ms/batch  %.0f slices/s" % (b, size, size, ms, b / ms * 1e3))
