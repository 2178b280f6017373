"""Plain-PyTorch-CPU restatement of the SAUNet hot path -- TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  This is the checker the HIP path is compared with; it is
never imported by the product package.

Written functionally over a flat ``{key: tensor}`` state dict that uses the
reference's state-dict key names, so the same weights drive (a) the real reference
(in the build container, via oracle/ref_import.py), (b) this restatement and (c) the
HIP modules.  Pinned against the real reference by tests/golden/*.npz
(oracle/make_golden.py) and, in the container, by tests/test_oracle_vs_reference.py.

Reference citations (all under /root/reference):
  SAUNet.forward ............ models/models.py:326-394
  DenseNet-121 slicing ...... models/models.py:304-313 (norm0 output used raw: no relu0/pool0)
  DualAttBlock .............. models/attention_blocks.py:208-238 (+ _MRF :175-206,
                              SpatialAttentionBlock :145-173, SEModule :28-57)
  GatedSpatialConv2d ........ models/GSConv.py:16-62
  BasicBlock (SyncBN) ....... models/resnet.py:30-59, lib/nn/modules/batchnorm.py:38-61
                              (momentum 0.001; single-device path is F.batch_norm)
  DecoderBlock .............. models/models.py:203-237
  DualLoss / dice_loss ...... loss.py:124-159 / :51-88
  pixel_acc ................. models/models.py:51-74
  SegmentationModule ........ models/models.py:80-109
  mask_to_edges ............. data/ac17_dataloader.py:231-258
  intersectionAndUnion ...... utils.py:119-140
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import canny as _canny

DENSE_BLOCKS = (6, 12, 24, 16)
BN_EPS = 1e-5
BN_MOM = 0.1
SYNCBN_MOM = 0.001  # lib/nn/modules/batchnorm.py:39
CE_WEIGHT = (1.0, 4.0, 5.0, 1.0)  # loss.py:130


# ----------------------------------------------------------------------------- primitives
# Data-parallel emulation (/root/reference/train.py:272-277, lib/nn/modules/batchnorm.py:98-139): when
# _DP_SHARDS = K > 1 the batch dimension holds K equal shards back to back (one per replica).  Convolutions, pooling, resampling
# and the SE pool are per-sample, so running them on the concatenation IS running them per replica; only batch normalisation and
# the loss see the shard structure:
#   * the 144 nn.BatchNorm2d layers normalise every shard with ITS OWN batch statistics (what nn.DataParallel replicas do);
#     the running statistics that survive are replica 0's (replica 0 shares the module's buffers, the others are discarded);
#   * the 6 SynchronizedBatchNorm2d layers (momentum == SYNCBN_MOM) use the statistics of the whole global batch, computed
#     and accumulated exactly like _compute_mean_std (clamp(var, eps)^-1/2, _tmp_running_* / _running_iter);
#   * the loss is evaluated per shard and averaged (train.py:96 `loss.mean()`).
_DP_SHARDS = 1

# bf16-storage emulation (test infrastructure for the bf16 parity tests): with _BF16_EMU set, every tensor an op hands to the next op
# -- activations on the way forward, their gradients on the way back -- and every convolution weight operand is rounded to bfloat16,
# while all arithmetic stays float32.  This is what ANY implementation that keeps activations / gradients / MFMA operands in bf16
# (the HIP path in its bench configuration, or the reference under bf16 autocast) computes, up to where exactly it rounds; the one- and
# two-channel edge head and the loss stay float32 like in the HIP path.  It separates "error inherent to bf16 storage" from
# "error of the implementation" in tests/test_hip_parity_bf16.py.
_BF16_EMU = False
# mixed-storage study: with _BF16_EMU_F32_MAXHW = m > 0 every tensor on a map of at most m x m pixels (and the weight
# operand of every convolution whose INPUT map is that small) stays float32 -- "float32 storage on the low-resolution maps, bf16 elsewhere"
_BF16_EMU_F32_MAXHW = 0


class bf16_storage:
    def __init__(self, on=True, f32_max_hw=0):
        self.on = bool(on)
        self.f32_max_hw = int(f32_max_hw)

    def __enter__(self):
        global _BF16_EMU, _BF16_EMU_F32_MAXHW
        self.prev, _BF16_EMU = (_BF16_EMU, _BF16_EMU_F32_MAXHW), self.on
        _BF16_EMU_F32_MAXHW = self.f32_max_hw
        return self

    def __exit__(self, *exc):
        global _BF16_EMU, _BF16_EMU_F32_MAXHW
        _BF16_EMU, _BF16_EMU_F32_MAXHW = self.prev


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _small(x):
    return _BF16_EMU_F32_MAXHW > 0 and x.dim() == 4 and max(x.shape[-2], x.shape[-1]) <= _BF16_EMU_F32_MAXHW


def _q(x):
    """stored activation: value and gradient live in bf16 (float32 on the small maps of the mixed-storage study)"""
    return _RoundBoth.apply(x) if (_BF16_EMU and not _small(x)) else x


def _qw(w, x=None):
    """weight operand: rounded for the multiply, gradient kept in float32 (master weights are float32); x = the convolution's input"""
    return _RoundFwd.apply(w) if (_BF16_EMU and not (x is not None and _small(x))) else w


class dp_shards:
    """context manager: interpret the batch as K per-replica shards"""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        global _DP_SHARDS
        self.prev, _DP_SHARDS = _DP_SHARDS, self.k
        return self

    def __exit__(self, *exc):
        global _DP_SHARDS
        _DP_SHARDS = self.prev


def _sync_bn_train(sd, pre, x, momentum):
    """_SynchronizedBatchNorm.forward, parallel branch (lib/nn/modules/batchnorm.py:63-84, 118-139)."""
    n = x.numel() // x.shape[1]
    s1 = x.sum((0, 2, 3)); s2 = (x * x).sum((0, 2, 3))
    mean = s1 / n
    sumvar = s2 - s1 * mean
    unbias_var, bias_var = sumvar / (n - 1), sumvar / n
    keep = 1.0 - momentum
    with torch.no_grad():
        tm = sd[pre + "._tmp_running_mean"] * keep + mean.detach()
        tv = sd[pre + "._tmp_running_var"] * keep + unbias_var.detach()
        it = sd[pre + "._running_iter"] * keep + 1
        sd[pre + "._tmp_running_mean"], sd[pre + "._tmp_running_var"], sd[pre + "._running_iter"] = tm, tv, it
        sd[pre + ".running_mean"], sd[pre + ".running_var"] = tm / it, tv / it
    inv_std = bias_var.clamp(BN_EPS) ** -0.5
    return (x - mean[None, :, None, None]) * (inv_std * sd[pre + ".weight"])[None, :, None, None] + sd[pre + ".bias"][None, :, None, None]


def _bn(sd, pre, x, training, momentum=BN_MOM):
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if training and (pre + ".num_batches_tracked") in sd:
        sd[pre + ".num_batches_tracked"] += 1
    k = _DP_SHARDS
    if k > 1 and training:
        if momentum == SYNCBN_MOM and (pre + "._running_iter") in sd:
            return _q(_sync_bn_train(sd, pre, x, momentum))
        b = x.shape[0] // k
        outs = []
        for r in range(k):       # replica r: local statistics; only replica 0's running-statistic update survives
            outs.append(F.batch_norm(x[r * b:(r + 1) * b], rm if r == 0 else rm.clone(), rv if r == 0 else rv.clone(),
                                     sd[pre + ".weight"], sd[pre + ".bias"], True, momentum, BN_EPS))
        return _q(torch.cat(outs, 0))
    return _q(F.batch_norm(x, rm, rv, sd[pre + ".weight"], sd[pre + ".bias"], training, momentum, BN_EPS))


def _conv(sd, pre, x, stride=1, padding=0):
    return _q(F.conv2d(x, _qw(sd[pre + ".weight"], x), sd.get(pre + ".bias"), stride, padding))


def _convT(sd, pre, x):
    return _q(F.conv_transpose2d(x, _qw(sd[pre + ".weight"], x), sd.get(pre + ".bias"), stride=2, padding=1))


def _up(x, size=None, scale=None):
    return _q(F.interpolate(x, size=size, scale_factor=scale, mode="bilinear", align_corners=True))


# ----------------------------------------------------------------------------- encoder
def dense_block(sd, pre, x, nlayers, training):
    feats = x
    for i in range(1, nlayers + 1):
        p = "%s.denselayer%d" % (pre, i)
        y = _conv(sd, p + ".conv1", F.relu(_bn(sd, p + ".norm1", feats, training)))
        y = _conv(sd, p + ".conv2", F.relu(_bn(sd, p + ".norm2", y, training)), padding=1)
        feats = torch.cat([feats, y], 1)
    return feats


def transition(sd, pre, x, training):
    return _q(F.avg_pool2d(_conv(sd, pre + ".conv", F.relu(_bn(sd, pre + ".norm", x, training))), 2, 2))


def encoder(sd, x, training, pre="encoder.features"):
    conv1 = _bn(sd, pre + ".norm0", _conv(sd, pre + ".conv0", x, stride=2, padding=3), training)
    conv2 = transition(sd, pre + ".transition1", dense_block(sd, pre + ".denseblock1", conv1, 6, training), training)
    conv3 = transition(sd, pre + ".transition2", dense_block(sd, pre + ".denseblock2", conv2, 12, training), training)
    conv4 = transition(sd, pre + ".transition3", dense_block(sd, pre + ".denseblock3", conv3, 24, training), training)
    conv5 = _bn(sd, pre + ".norm5", dense_block(sd, pre + ".denseblock4", conv4, 16, training), training)
    return conv1, conv2, conv3, conv4, conv5


# ----------------------------------------------------------------------------- blocks
def basic_block(sd, pre, x, training):
    y = F.relu(_bn(sd, pre + ".bn1", _conv(sd, pre + ".conv1", x, padding=1), training, SYNCBN_MOM))
    y = _bn(sd, pre + ".bn2", _conv(sd, pre + ".conv2", y, padding=1), training, SYNCBN_MOM)
    return _q(F.relu(y + x))


def gated_conv(sd, pre, feat, gate, training):
    g = pre + "._gate_conv"
    a = _bn(sd, g + ".0", torch.cat([feat, gate], 1), training)
    a = F.relu(_conv(sd, g + ".1", a))
    a = torch.sigmoid(_bn(sd, g + ".4", _conv(sd, g + ".3", a), training))
    return _q(F.conv2d(_q(feat * (a + 1)), _qw(sd[pre + ".weight"]), sd.get(pre + ".bias"))), a


def conv3x3_bn_relu(sd, pre, x, training):
    return F.relu(_bn(sd, pre + ".1", _conv(sd, pre + ".0", x, padding=1), training))


def se_module(sd, pre, x):
    s = F.adaptive_avg_pool2d(x, 1)
    s = torch.sigmoid(_conv(sd, pre + ".fc2", F.relu(_conv(sd, pre + ".fc1", s))))
    return x * s


def spatial_attention(sd, pre, x, training):
    c = F.relu(_bn(sd, pre + ".bn", _conv(sd, pre + ".down", x), training))
    return torch.sigmoid(_conv(sd, pre + ".phi", c))


def dual_att_block(sd, pre, low, skip, training):
    up = F.relu(_bn(sd, pre + ".mrf.up.1", _convT(sd, pre + ".mrf.up.0", low), training))
    fused = conv3x3_bn_relu(sd, pre + ".c3x3rb", torch.cat([skip, up], 1), training)
    spatial = spatial_attention(sd, pre + ".spatialAttn", fused, training)
    channel = se_module(sd, pre + ".channelAttn", fused)
    return _q((spatial + 1) * channel), spatial


def decoder_block(sd, pre, x, training):
    y = conv3x3_bn_relu(sd, pre + ".block.0", x, training)
    return F.relu(_bn(sd, pre + ".block.2", _convT(sd, pre + ".block.1", y), training))


def canny_branch(x):
    """models/models.py:359-363 -- host-side, no gradient, values {0,255}."""
    return torch.from_numpy(_canny.canny_batch(x.detach().cpu().numpy()))


# ----------------------------------------------------------------------------- whole net
def saunet_forward(sd, x, training=True, return_att=False, canny=None):
    size = x.shape[2:]
    conv1, conv2, conv3, conv4, conv5 = encoder(sd, x, training)

    ss = basic_block(sd, "res1", _up(_conv(sd, "d0", conv2), size=size), training)
    c3 = _up(_conv(sd, "c3", conv3), size=size)
    ss, g1 = gated_conv(sd, "gate1", _conv(sd, "d1", ss), c3, training)
    ss = _conv(sd, "d2", basic_block(sd, "res2", ss, training))
    c4 = _up(_conv(sd, "c4", conv4), size=size)
    ss, g2 = gated_conv(sd, "gate2", ss, c4, training)
    ss = _conv(sd, "d3", basic_block(sd, "res3", ss, training))
    c5 = _up(_conv(sd, "c5", conv5), size=size)
    ss, g3 = gated_conv(sd, "gate3", ss, c5, training)
    ss = _up(_conv(sd, "fuse", ss), size=size)
    edge_out = torch.sigmoid(ss)

    if canny is None:
        canny = canny_branch(x)
    with bf16_storage(False):        # the one/two-channel edge head is float32 in every storage mode (saunet_amd/modules.py: SAUNet.forward)
        acts = torch.sigmoid(_conv(sd, "cw", torch.cat([edge_out, canny.to(edge_out.dtype)], 1)))
        edge = _bn(sd, "expand.1", _conv(sd, "expand.0", acts), training)
    edge = _q(F.relu(edge))

    conv2u, conv3u, conv4u = _up(conv2, scale=2), _up(conv3, scale=2), _up(conv4, scale=2)
    center = conv3x3_bn_relu(sd, "center", _q(F.max_pool2d(conv5, 2, 2)), training)
    dec5, att5 = dual_att_block(sd, "dec5", center, conv5, training)
    dec4, att4 = dual_att_block(sd, "dec4", dec5, conv4u, training)
    dec3, att3 = dual_att_block(sd, "dec3", dec4, conv3u, training)
    dec2, att2 = dual_att_block(sd, "dec2", dec3, conv2u, training)
    dec1 = decoder_block(sd, "dec1", dec2, training)
    dec0 = conv3x3_bn_relu(sd, "dec0", torch.cat([dec1, edge], 1), training)
    logits = _conv(sd, "final", dec0)
    if return_att:
        maps = [_up(att2, scale=2), _up(att3, scale=4), _up(att4, scale=8), _up(att5, scale=16), g1, g2, g3]
        return logits, edge_out, maps
    return logits, edge_out


# ----------------------------------------------------------------------------- loss / metrics
def dice_loss(seg_t, logits, eps=1e-7):
    onehot = F.one_hot(seg_t.long(), logits.shape[1]).permute(0, 3, 1, 2).to(logits.dtype)
    p = F.softmax(logits, 1)
    inter = (p * onehot).sum((0, 2, 3))
    card = (p + onehot).sum((0, 2, 3))
    return 1 - (2.0 * inter / (card + eps)).mean()


def dual_loss(logits, edge_prob, seg_t, edge_t):
    w = torch.tensor(CE_WEIGHT, dtype=logits.dtype)
    ce = F.cross_entropy(logits, seg_t.long(), weight=w)
    bce = F.binary_cross_entropy(edge_prob, edge_t.to(edge_prob.dtype))
    return dice_loss(seg_t, logits) + ce + bce


def pixel_acc(logits, seg_t, num_class=4):
    pred = torch.round(F.softmax(logits, 1)).long().argmax(1)  # torch.max ties -> lowest index
    lab = seg_t.long()
    valid = lab >= 1
    acc = (valid & (pred == lab)).sum().float() / (valid.sum().float() + 1e-10)
    jac = []
    for c in range(1, num_class):
        v, q = lab == c, pred == c
        anb = (v & q).sum().float()
        jac.append(anb / (v.sum().float() + q.sum().float() - anb + 1e-10))
    return acc, jac


def segmentation_step(sd, image, seg_t, edge_t, training=True, canny=None):
    """SegmentationModule.forward train branch (models/models.py:89-93)."""
    logits, edge = saunet_forward(sd, image, training, canny=canny)
    loss = dual_loss(logits, edge, seg_t, edge_t)
    acc = pixel_acc(logits.detach(), seg_t)
    return loss, acc, logits, edge


def dp_emulate_step(sd, shards, training=True):
    """K-replica data-parallel step on the CPU: `shards` = [(image, seg, edge)] * K with identical
    weights `sd`; returns (mean of the per-shard losses, [per-shard loss], logits, edge) -- backpropagating the mean loss
    yields the average of the K replica gradients, which is what the all-reduce delivers to every rank."""
    k = len(shards)
    b = shards[0][0].shape[0]
    assert all(s[0].shape[0] == b for s in shards), "equal shards (drop_last=True, train.py:252)"
    image = torch.cat([s[0] for s in shards]); seg_t = torch.cat([s[1] for s in shards]); edge_t = torch.cat([s[2] for s in shards])
    # SyncBN bookkeeping buffers as the reference CONSTRUCTOR leaves them (batchnorm.py:50-54: running_mean = 0, running_var = 1,
    # iter = 1 at that point) when the state dict lacks them -- what load_state_dict(strict=False) of such a checkpoint gives
    for key in list(sd.keys()):
        if key.endswith(".running_mean") and key.split(".")[0] in ("res1", "res2", "res3"):
            pre = key[:-len(".running_mean")]
            sd.setdefault(pre + "._running_iter", torch.ones(1))
            sd.setdefault(pre + "._tmp_running_mean", torch.zeros_like(sd[key]))
            sd.setdefault(pre + "._tmp_running_var", torch.ones_like(sd[key]))
    with dp_shards(k):
        logits, edge = saunet_forward(sd, image, training)
    losses = [dual_loss(logits[r * b:(r + 1) * b], edge[r * b:(r + 1) * b], seg_t[r * b:(r + 1) * b], edge_t[r * b:(r + 1) * b])
              for r in range(k)]
    return torch.stack(losses).mean(), losses, logits, edge


def intersection_and_union(pred, lab, num_class):
    pred = np.asarray(pred).copy() + 1
    lab = np.asarray(lab).copy() + 1
    pred = pred * (lab > 0)
    inter = pred * (pred == lab)
    ai, _ = np.histogram(inter, bins=num_class, range=(1, num_class))
    ap, _ = np.histogram(pred, bins=num_class, range=(1, num_class))
    al, _ = np.histogram(lab, bins=num_class, range=(1, num_class))
    return ai, ap + al - ai


def mask_to_edges(mask, radius=2, num_classes=3):
    """mask: int [H,W] in {0..3} -> float32 [1,H,W] in {0,1}."""
    from scipy.ndimage import distance_transform_edt
    mask = np.asarray(mask)
    edge = np.zeros(mask.shape, dtype=np.float64)
    for c in range(1, num_classes + 1):
        m = np.pad((mask == c), 1, mode="constant").astype(np.float64)
        d = (distance_transform_edt(m) + distance_transform_edt(1.0 - m))[1:-1, 1:-1]
        d[d > radius] = 0
        edge += d
    return (edge > 0).astype(np.float32)[None]


# ----------------------------------------------------------------------------- state dict
def state_dict_spec(num_classes=4, nf=32):
    """[(key, shape, kind)] for every tensor in the reference SAUNet state dict that the
    forward pass reads (aliases conv1.*..conv5.* and the unused classifier omitted)."""
    spec = []

    def conv(k, co, ci, kh, bias):
        spec.append((k + ".weight", (co, ci, kh, kh), "conv"))
        if bias:
            spec.append((k + ".bias", (co,), "bias"))

    def convT(k, ci, co, bias=True):
        spec.append((k + ".weight", (ci, co, 4, 4), "conv"))
        if bias:
            spec.append((k + ".bias", (co,), "bias"))

    def bn(k, c):
        spec.extend([(k + ".weight", (c,), "gamma"), (k + ".bias", (c,), "beta"),
                     (k + ".running_mean", (c,), "rmean"), (k + ".running_var", (c,), "rvar"),
                     (k + ".num_batches_tracked", (), "count")])

    e = "encoder.features"
    conv(e + ".conv0", 64, 3, 7, False)
    bn(e + ".norm0", 64)
    c = 64
    for b, n in enumerate(DENSE_BLOCKS, 1):
        for i in range(1, n + 1):
            p = "%s.denseblock%d.denselayer%d" % (e, b, i)
            bn(p + ".norm1", c)
            conv(p + ".conv1", 128, c, 1, False)
            bn(p + ".norm2", 128)
            conv(p + ".conv2", 32, 128, 3, False)
            c += 32
        if b < 4:
            bn("%s.transition%d.norm" % (e, b), c)
            conv("%s.transition%d.conv" % (e, b), c // 2, c, 1, False)
            c //= 2
    bn(e + ".norm5", c)
    conv("c3", 1, 256, 1, True); conv("c4", 1, 512, 1, True); conv("c5", 1, 1024, 1, True)
    conv("d0", 64, 128, 1, True); conv("d1", 32, 64, 1, True); conv("d2", 16, 32, 1, True); conv("d3", 8, 16, 1, True)
    conv("fuse", 1, 8, 1, False); conv("cw", 1, 2, 1, False)
    for k, ch in (("res1", 64), ("res2", 32), ("res3", 16)):
        conv(k + ".conv1", ch, ch, 3, False); bn(k + ".bn1", ch)
        conv(k + ".conv2", ch, ch, 3, False); bn(k + ".bn2", ch)
    for k, ch in (("gate1", 32), ("gate2", 16), ("gate3", 8)):
        spec.append((k + ".weight", (ch, ch, 1, 1), "conv"))
        bn(k + "._gate_conv.0", ch + 1)
        conv(k + "._gate_conv.1", ch + 1, ch + 1, 1, True)
        conv(k + "._gate_conv.3", 1, ch + 1, 1, True)
        bn(k + "._gate_conv.4", 1)
    conv("expand.0", nf, 1, 1, True); bn("expand.1", nf)
    conv("center.0", nf * 16, 1024, 3, True); bn("center.1", nf * 16)
    for k, (lo, sk, co) in (("dec5", (512, 1024, 512)), ("dec4", (512, 512, 256)),
                            ("dec3", (256, 256, 128)), ("dec2", (128, 128, 64))):
        convT(k + ".mrf.up.0", lo, lo); bn(k + ".mrf.up.1", lo)
        conv(k + ".c3x3rb.0", co, lo + sk, 3, True); bn(k + ".c3x3rb.1", co)
        conv(k + ".spatialAttn.down", co // 4, co, 1, False)
        conv(k + ".spatialAttn.phi", 1, co // 4, 1, True)
        bn(k + ".spatialAttn.bn", co // 4)
        conv(k + ".channelAttn.fc1", co // 16, co, 1, True)
        conv(k + ".channelAttn.fc2", co, co // 16, 1, True)
    conv("dec1.block.0.0", 48, 64, 3, True); bn("dec1.block.0.1", 48)
    convT("dec1.block.1", 48, nf); bn("dec1.block.2", nf)
    conv("dec0.0", nf, nf * 2, 3, True); bn("dec0.1", nf)
    conv("final", num_classes, nf, 1, True)
    return spec
