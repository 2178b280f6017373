"""Generate tests/golden/*.npz by running the REAL reference -- CONTAINER-ONLY.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Imports /root/reference through oracle/ref_import.py (stand-ins for torchvision / cv2,
.cuda() -> identity), drives its modules on seeded inputs with weights from
oracle/weights.py and stores inputs-by-seed + expected outputs / gradients /
running statistics.  The fixtures are data only; no reference source travels.

Fixtures
--------
modules_*.npz   one per reference module (train mode: out, dX, dW..., new running stats;
                eval mode: out)
loss.npz        dice_loss / DualLoss / pixel_acc / intersectionAndUnion / mask_to_edges
saunet_128.npz  config #1 (B=2, 128x128): loss, metrics, 8x-subsampled logits/edge,
                per-parameter gradient L2 norms, 10-step SGD loss trajectory, eval branch
canny_cast.npz  the float->uint8 cast observed on this x86 numpy (models.py:359)
"""
import contextlib
import io
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, saunet_ref as R, weights as Wt  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def module_kinds(mod):
    kinds = {}
    for mname, m in mod.named_modules():
        pre = mname + "." if mname else ""
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            kinds[pre + "weight"] = "gamma"; kinds[pre + "bias"] = "beta"
            kinds[pre + "running_mean"] = "rmean"; kinds[pre + "running_var"] = "rvar"
        elif isinstance(m, nn.modules.conv._ConvNd):
            kinds[pre + "weight"] = "conv"
            if m.bias is not None:
                kinds[pre + "bias"] = "bias"
    return kinds


def fill_module(mod, tag, seed):
    kinds = module_kinds(mod)
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        if k in kinds:
            new[k] = Wt.make_tensor(tag + "." + k, tuple(v.shape), kinds[k], seed)
    mod.load_state_dict(new, strict=False)
    return kinds


def rnd(shape, seed, key, scale=1.0):
    r = np.random.default_rng([seed, zlib.crc32(key.encode())])
    return torch.from_numpy((r.standard_normal(shape).astype(np.float32) * np.float32(scale)))


def run_module(name, mod, inputs, seed, call=None):
    """train fwd/bwd + eval fwd; returns flat dict of arrays."""
    out = {}
    kinds = fill_module(mod, name, seed)
    sd0 = {k: v.clone() for k, v in mod.state_dict().items()}
    mod.train()
    xs = [x.clone().requires_grad_(True) for x in inputs]
    y = call(mod, xs) if call else mod(*xs)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    cots = [rnd(tuple(t.shape), seed, "%s.cot%d" % (name, i)) for i, t in enumerate(ys)]
    torch.autograd.backward(ys, cots)
    for i, t in enumerate(ys):
        out["train.out%d" % i] = t.detach().numpy()
    for i, x in enumerate(xs):
        out["train.dx%d" % i] = x.grad.numpy()
    for k, p in mod.named_parameters():
        out["train.grad." + k] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    for k, b in mod.named_buffers():
        if kinds.get(k) in ("rmean", "rvar"):
            out["train.buf." + k] = b.detach().numpy().copy()
    mod.load_state_dict(sd0)
    mod.eval()
    with torch.no_grad():
        y = call(mod, [x.detach() for x in inputs]) if call else mod(*[x.detach() for x in inputs])
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    for i, t in enumerate(ys):
        out["eval.out%d" % i] = t.numpy()
    return out


def gen_modules(ns, seed=7):
    B, H = 2, 8
    specs = {
        # name: (ctor, input shapes, call)
        "SEModule": (lambda: ns.SEModule(32, 16), [(B, 32, H, H)], None),
        "SpatialAttentionBlock": (lambda: ns.SpatialAttentionBlock(32, 8, 2), [(B, 32, H, H)], None),
        "DualAttBlock": (lambda: ns.DualAttBlock(inchannels=[32, 48], outchannels=32),
                         [(B, 32, H // 2, H // 2), (B, 48, H, H)], lambda m, xs: m([xs[0], xs[1]])),
        "GatedSpatialConv2d": (lambda: ns.GatedSpatialConv2d(16, 16), [(B, 16, H, H), (B, 1, H, H)], None),
        "BasicBlock": (lambda: ns.BasicBlock(16, 16), [(B, 16, H, H)], None),
        "DecoderBlock": (lambda: ns.DecoderBlock(32, 24, 16, True), [(B, 32, H, H)], None),
        "conv3x3_bn_relu": (lambda: ns.conv3x3_bn_relu(24, 16), [(B, 24, H, H)], None),
    }
    for name, (ctor, shapes, call) in specs.items():
        with contextlib.redirect_stdout(io.StringIO()):
            mod = ctor()
        inputs = [rnd(s, seed, "%s.in%d" % (name, i)) for i, s in enumerate(shapes)]
        out = run_module(name, mod, inputs, seed, call)
        out["meta.seed"] = np.int64(seed)
        for i, s in enumerate(shapes):
            out["meta.shape%d" % i] = np.array(s, np.int64)
        np.savez_compressed(os.path.join(GOLD, "modules_%s.npz" % name), **out)
        print("wrote modules_%s.npz (%d arrays)" % (name, len(out)))


def gen_loss(ns, seed=11):
    B, H = 3, 16
    logits = rnd((B, 4, H, H), seed, "loss.logits", 2.0).requires_grad_(True)
    edge = torch.sigmoid(rnd((B, 1, H, H), seed, "loss.edge", 2.0)).requires_grad_(True)
    r = np.random.default_rng(seed)
    seg = torch.from_numpy(r.integers(0, 4, (B, H, H)))
    edge_t = torch.from_numpy((r.random((B, 1, H, H)) < 0.2).astype(np.float32))
    out = {"seg": seg.numpy(), "edge_t": edge_t.numpy()}
    d = ns.dice_loss(seg, logits)
    out["dice"] = d.detach().numpy()
    crit = ns.DualLoss(mode="train")
    L = crit((logits, edge), (seg.double(), edge_t), epoch=3)
    L.backward()
    out["dual"] = L.detach().numpy()
    out["dlogits"] = logits.grad.numpy(); out["dedge"] = edge.grad.numpy()
    base = ns.models.SegmentationModuleBase()
    acc, jac = base.pixel_acc(torch.round(torch.softmax(logits.detach(), 1)).long(), seg, 4)
    out["acc"] = acc.numpy(); out["jac"] = np.array([float(j) for j in jac], np.float32)
    pred = logits.detach().argmax(1).numpy()
    ai, au = ns.intersectionAndUnion(pred[0], seg.numpy()[0], 4)
    out["iau_i"] = ai; out["iau_u"] = au
    # mask_to_edges: call the reference method unbound on a phantom label map
    _, seg_ph, _ = Wt.synthetic_batch(1, 48, 48, seed=5)
    if ns.AC17_2DLoad is not None:
        holder = ns.AC17_2DLoad.__new__(ns.AC17_2DLoad)
        e = ns.AC17_2DLoad.mask_to_edges(holder, seg_ph[0].numpy())
        out["m2e_mask"] = seg_ph[0].numpy(); out["m2e_edge"] = e.numpy()
    out["meta.seed"] = np.int64(seed)
    np.savez_compressed(os.path.join(GOLD, "loss.npz"), **out)
    print("wrote loss.npz")


def gen_cast():
    v = np.array([-2.3, -1.0, -0.4, 0.6, 1.7, 3.9, -0.999, -255.5, -256.0, -300.7, 254.9, 255.0, 256.2, 700.1,
                  -1e-3, 12.5], np.float32)
    np.savez_compressed(os.path.join(GOLD, "canny_cast.npz"), v=v, u8=v.astype(np.uint8))
    print("wrote canny_cast.npz", v.astype(np.uint8))


def gen_saunet(ns, seed=3, B=2, H=128):
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, seed)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ns.SAUNet(num_classes=4)
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    img, seg, edge = Wt.synthetic_batch(B, H, H)
    sm = ns.SegmentationModule(ns.DualLoss(mode="train"), net, 4)
    sm.train()
    out = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.H": np.int64(H)}
    loss, (acc, jac) = sm({"image": img, "mask": (seg.double(), edge)}, 1)
    loss.backward()
    out["loss0"] = loss.detach().numpy(); out["acc0"] = acc.numpy()
    out["jac0"] = np.array([float(j) for j in jac], np.float32)
    pd = dict(net.named_parameters())
    keys = Wt.trainable_keys(spec)
    out["grad_keys"] = np.array(keys)
    out["grad_norms"] = np.array([float(pd[k].grad.double().norm()) for k in keys], np.float64)
    out["grad_sums"] = np.array([float(pd[k].grad.double().sum()) for k in keys], np.float64)
    # a few full gradients of small tensors
    for k in ("final.weight", "final.bias", "cw.weight", "fuse.weight", "gate1.weight", "d3.weight",
              "encoder.features.denseblock4.denselayer16.norm2.weight", "res1.bn1.weight", "c5.weight",
              "dec2.spatialAttn.phi.weight", "dec5.channelAttn.fc1.bias"):
        out["grad." + k] = pd[k].grad.numpy()
    bd = dict(net.named_buffers())
    for k in ("res1.bn1.running_mean", "res1.bn1.running_var", "encoder.features.norm5.running_var",
              "dec0.1.running_mean", "gate3._gate_conv.4.running_var"):
        out["buf." + k] = bd[k].numpy().copy()
    # forward tensors (fresh weights, train mode) subsampled 8x
    net.load_state_dict(sd, strict=False); net.zero_grad()
    with torch.no_grad():
        lg, eo = net(img)
    out["logits_s8"] = lg[:, :, ::8, ::8].numpy(); out["edge_s8"] = eo[:, :, ::8, ::8].numpy()
    out["logits_sum"] = np.float64(lg.double().sum()); out["edge_sum"] = np.float64(eo.double().sum())
    # inference branch (models/models.py:105-109) with the INITIAL weights, eval-mode BN (running stats)
    net.load_state_dict(sd, strict=False)
    sm.eval()
    with torch.no_grad():
        pred0, l_eval0 = sm({"image": img[:1], "mask": (seg[0], edge[0])}, epoch=0, segSize=(H, H))
    out["eval0_pred_s8"] = pred0[:, :, ::8, ::8].numpy(); out["eval0_loss"] = l_eval0.numpy()
    sm.train()
    # 10 SGD steps (config #1: lr 5e-4, m 0.9, wd 1e-4 on conv/linear weights only; train.py:166-196)
    net.load_state_dict(sd, strict=False); net.zero_grad()
    decay, no_decay = [], []
    for m in net.modules():
        if isinstance(m, nn.Linear) or isinstance(m, nn.modules.conv._ConvNd):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            no_decay += [m.weight, m.bias]
    opt = torch.optim.SGD([dict(params=decay), dict(params=no_decay, weight_decay=0.0)], lr=5e-4, momentum=0.9,
                          weight_decay=1e-4, nesterov=False)
    traj = []
    for it in range(10):
        sm.zero_grad()
        loss, _ = sm({"image": img, "mask": (seg.double(), edge)}, 1)
        loss.backward(); opt.step(); traj.append(float(loss))
    out["sgd_traj"] = np.array(traj, np.float64)
    # eval / inference branch (models/models.py:105-109) on sample 0 with the trained weights
    sm.eval()
    with torch.no_grad():
        pred, l_eval = sm({"image": img[:1], "mask": (seg[0], edge[0])}, epoch=0, segSize=(H, H))
    out["eval_pred_s8"] = pred[:, :, ::8, ::8].numpy(); out["eval_loss"] = l_eval.numpy()
    out["eval_pred_argmax_hist"] = np.bincount(pred.argmax(1).flatten().numpy(), minlength=4)
    np.savez_compressed(os.path.join(GOLD, "saunet_128.npz"), **out)
    print("wrote saunet_128.npz; loss0=%.6f traj=%s" % (float(out["loss0"]), traj))


if __name__ == "__main__":
    torch.set_num_threads(8)
    torch.manual_seed(0)
    os.makedirs(GOLD, exist_ok=True)
    ns = ref_import.load()
    gen_cast()
    gen_modules(ns)
    gen_loss(ns)
    gen_saunet(ns)
