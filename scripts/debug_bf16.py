import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import saunet_amd as S
from saunet_amd import functional as HF
from oracle import saunet_ref as R, weights as Wt
import torch.nn.functional as F

dt = torch.bfloat16 if len(sys.argv) < 2 else getattr(torch, sys.argv[1])
S.set_compute_dtype(dt)
spec = R.state_dict_spec(); sd = Wt.make_state_dict(spec, 3)
net = S.SAUNet(num_classes=4).cuda(); net.load_state_dict(sd, strict=False); net.train()
img, seg, edge = Wt.synthetic_batch(2, 128, 128)
sdo = {k: v.clone() for k, v in sd.items()}
with torch.no_grad():
    c1, c2, c3, c4, c5 = R.encoder(sdo, img, True)
    x = img.cuda()
    d1 = net.conv1(net._prep_input(x))
    def rel(a, b, name):
        a = a.float().cpu(); 
        print("%-10s rel err %.4f  (max ref %.3f, max dev %.3f)" % (name, float((a-b).abs().max()/b.abs().max()), float(b.abs().max()), float(a.abs().max())))
    rel(d1, c1, "conv1")
    buf, st = net.conv2(d1, with_stats=True)
    ob = R.dense_block(sdo, "encoder.features.denseblock1", c1, 6, True)
    rel(buf, ob, "block1")
    for c in range(64, 256, 32):
        rel(buf[:, c:c+32], ob[:, c:c+32], " ch%d" % c)
    d2 = net.conv2t(buf, st); rel(d2, c2, "conv2")
    buf, st = net.conv3(d2, with_stats=True); d3 = net.conv3t(buf, st); rel(d3, c3, "conv3")
    buf, st = net.conv4(d3, with_stats=True); d4 = net.conv4t(buf, st); rel(d4, c4, "conv4")
    d5 = net.conv5(d4); rel(d5, c5, "conv5")
    lg_o, eo_o = R.saunet_forward(sdo, img, True)
    net.load_state_dict(sd, strict=False)
    lg, eo = net(x)
    rel(lg, lg_o, "logits"); rel(eo, eo_o, "edge_out")
    print("edge_out dtype", eo.dtype, "min/max", float(eo.min()), float(eo.max()), " oracle", float(eo_o.min()), float(eo_o.max()))
    print("oracle loss on oracle outs", float(R.dual_loss(lg_o, eo_o, seg, edge)))
    print("oracle loss on device outs", float(R.dual_loss(lg.float().cpu(), eo.float().cpu(), seg, edge)))
    loss, m = HF.dual_loss(lg, eo, seg.cuda(), edge.cuda())
    print("device loss on device outs", float(loss))
    print("terms oracle: dice %.4f ce %.4f bce %.4f" % (float(R.dice_loss(seg, lg_o)), float(F.cross_entropy(lg_o, seg, weight=torch.tensor(R.CE_WEIGHT))), float(F.binary_cross_entropy(eo_o, edge))))
    l2, e2 = lg.float().cpu(), eo.float().cpu()
    print("terms device-outs: dice %.4f ce %.4f bce %.4f" % (float(R.dice_loss(seg, l2)), float(F.cross_entropy(l2, seg, weight=torch.tensor(R.CE_WEIGHT))), float(F.binary_cross_entropy(e2, edge))))
