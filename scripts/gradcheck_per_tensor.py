import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import saunet_ref as R, weights as Wt
import saunet_amd as S
for (B,H,W,seed) in [(2,128,128,0),(1,64,96,7),(4,256,256,13)]:
    S.set_compute_dtype(torch.float32)
    spec = R.state_dict_spec(); sd = Wt.make_state_dict(spec, seed)
    net = S.SAUNet(num_classes=4).cuda(); net.load_state_dict(sd, strict=False)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    img, seg, edge = Wt.synthetic_batch(B, H, W, seed=100+seed)
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = Wt.trainable_keys(spec)
    for k in keys: sdo[k].requires_grad_(True)
    lo, *_ = R.segmentation_step(sdo, img, seg, edge, True); lo.backward()
    loss, _ = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1); loss.backward()
    pd = dict(net.named_parameters())
    rows = []
    for k in keys:
        g, r = pd[k].grad.cpu().double(), sdo[k].grad.double()
        n = r.numel()
        rows.append((k, float((g-r).norm()/n**0.5), float(r.norm()/n**0.5), float((g-r).abs().max()), float(r.abs().max())))
    G = max(r[2] for r in rows); gmax = max(r[4] for r in rows)
    rel = sorted(((e/max(rr,1e-300), k, e, rr) for k,e,rr,_,_ in rows), reverse=True)
    print("case", B,H,W, "global rms scale %.3e gmax %.3e" % (G, gmax))
    for q in (1e-2,1e-3,1e-4,1e-5):
        big = [(x,k,e,rr) for x,k,e,rr in rel if rr >= q*G]
        print("  tensors with rms >= %.0e*G: %d, worst rel-L2 %.2e (%s)" % (q, len(big), big[0][0], big[0][1]))
    print("  worst 5 overall:", [(round(x,4), k, "%.1e"%rr) for x,k,e,rr in rel[:5]])
    print("  max e/G: %.2e" % max(e/G for _,k,e,rr in rel))
