import os, sys, subprocess, torch
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
w = os.path.join(HERE, "dp_worker.py")
env = dict(os.environ)
subprocess.run([sys.executable, w, "/tmp/s1.pt"], env=env, check=True)
subprocess.run([sys.executable, w, "/tmp/s2.pt"], env=env, check=True)
env2 = dict(env, SAUNET_DIST_BACKEND="gloo", SAUNET_SHARE_GPU="1")
subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534", w, "/tmp/d.pt"], env=env2, check=True, capture_output=True)
a, b, d = torch.load("/tmp/s1.pt"), torch.load("/tmp/s2.pt"), torch.load("/tmp/d.pt")
print("losses", a["losses"], b["losses"], d["losses"])
print("single vs single max diff", float((a["params"] - b["params"]).abs().max()))
diff = (a["params"] - d["params"]).abs()
print("single vs dp max diff", float(diff.max()), "at", int(diff.argmax()), "n big", int((diff > 1e-4).sum()))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import saunet_amd as S
net = S.SAUNet(num_classes=4)
o = 0
for n, p in net.named_parameters():
    k = p.numel()
    m = float(diff[o:o + k].max())
    if m > 1e-4: print("  %-60s %.3e" % (n, m))
    o += k
