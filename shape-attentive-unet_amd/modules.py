"""Drop-in ``torch.nn.Module`` replacements for the reference's SAUNet path, executing on libsaunet_hip.so.

Same class names, constructor arguments, forward signatures, state-dict keys and parameter discovery
(``isinstance`` of ``nn.modules.conv._ConvNd`` / ``nn.modules.batchnorm._BatchNorm`` as walked by
/root/reference/train.py:166-185) as the reference modules they replace:

  SAUNet, SegmentationModule, ModelBuilder, DecoderBlock, conv3x3_bn_relu .. models/models.py
  DualAttBlock, _MRF, SpatialAttentionBlock, SEModule ..................... models/attention_blocks.py
  GatedSpatialConv2d ...................................................... models/GSConv.py:16-62
  BasicBlock .............................................................. models/resnet.py:30-59
  Norm2d .................................................................. models/norm.py:16-22
  SynchronizedBatchNorm2d ................................................. lib/nn/modules/batchnorm.py:205-265
  DualLoss (dice_loss fused inside) ....................................... loss.py:124-159, :51-88
  DenseNet121 (torchvision layout) ........................................ third-party, models/models.py:271

``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` objects are used purely as parameter
containers (so checkpoints and optimiser grouping are interchangeable); their own forward is never called.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import functional as HF

_COMPUTE_DTYPE = torch.float32


def set_compute_dtype(dtype):
    """Storage dtype of activations / packed weights for modules built afterwards (float32 or bfloat16)."""
    global _COMPUTE_DTYPE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _COMPUTE_DTYPE = dtype


def get_compute_dtype():
    return _COMPUTE_DTYPE


def Norm2d(in_channels):
    return nn.BatchNorm2d(in_channels)


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    """Cross-replica BatchNorm of the shape-stream ResBlocks: one process per GPU, statistics all-reduced over
    RCCL inside conv_bn_act.  momentum defaults to 0.001 like the reference's vendored copy."""
    sync = True

    def __init__(self, num_features, eps=1e-5, momentum=0.001, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        # bookkeeping buffers of the reference implementation (kept for state-dict compatibility)
        self.register_buffer("_tmp_running_mean", torch.zeros(num_features))
        self.register_buffer("_tmp_running_var", torch.ones(num_features))
        self.register_buffer("_running_iter", torch.ones(1))

    def forward(self, x):
        return HF.batch_norm_act(x, self, relu=False)


def _init_conv_bn(module, conv_types=(nn.Conv2d, nn.ConvTranspose2d)):
    for m in module.modules():
        if isinstance(m, conv_types):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class ConvBNReLU(nn.Sequential):
    """conv3x3 -> BatchNorm -> ReLU as one fused unit (children '0', '1', '2' like the reference's nn.Sequential)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=True):
        super().__init__(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias),
                         nn.BatchNorm2d(out_planes), nn.ReLU(inplace=True))

    def forward(self, x, out_dtype=None, out=None, pool=None):
        """out: optional destination (a channel slice of a concat buffer, see HF.cat_alias); ignored where a cast is needed.
        pool: optional float32 [N, C] tensor receiving the global average pool of the result (SE squeeze fused into the BN-apply pass)."""
        conv, bn = self[0], self[1]
        if HF.expand_fusable(x, conv, bn):
            return HF.expand_bn_act(x, conv, bn, relu=True, out_dtype=out_dtype, out=out)      # 1 -> C broadcast (SAUNet.expand)
        direct = out is not None and (out_dtype is None or out_dtype == x.dtype)
        y = HF.conv_bn_act(x, conv.weight, conv.bias, bn, relu=True, stride=conv.stride[0], padding=conv.padding[0], out=out if direct else None,
                           pool=pool)
        if out_dtype is None or out_dtype == y.dtype:
            return y
        # a cast is needed (e.g. the float32 edge head feeding a bf16 decoder outside the fused `expand` kernel: eval mode with gradients, or
        # num_filters other than 32 / 64): the cast itself then writes the caller's destination slice, so cat_alias still finds the piece in place
        return HF.cast(y, out_dtype, out=out)


def conv3x3_bn_relu(in_planes, out_planes, stride=1):
    return ConvBNReLU(in_planes, out_planes, 3, stride, 1)


# ------------------------------------------------------------------------------------------------ shape stream
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        if stride != 1 or downsample is not None:
            raise NotImplementedError("SAUNet only uses stride-1 BasicBlocks without downsample")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = SynchronizedBatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = SynchronizedBatchNorm2d(planes)
        self.downsample = None
        self.stride = stride

    def forward(self, x):
        return HF.basic_block(x, self.conv1, self.bn1, self.conv2, self.bn2)


class GatedSpatialConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if self.kernel_size != (1, 1) or groups != 1:
            raise NotImplementedError("gate convolution is 1x1, groups=1 on this path")
        self._gate_conv = nn.Sequential(
            Norm2d(in_channels + 1), nn.Conv2d(in_channels + 1, in_channels + 1, 1), nn.ReLU(),
            nn.Conv2d(in_channels + 1, 1, 1), Norm2d(1), nn.Sigmoid())
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_normal_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, input_features, gating_features):
        g = self._gate_conv
        if HF.gated_conv_fusable(input_features, gating_features, self):
            return HF.gated_conv(input_features, gating_features, self)
        a = HF.batch_norm_act(HF.cat([input_features, gating_features]), g[0], relu=False)
        a = HF.relu(HF.conv2d(a, g[1].weight, g[1].bias))
        # conv (C+1 -> 1) -> BatchNorm(1) -> sigmoid
        a = HF.sigmoid(HF.conv_bn_act(a, g[3].weight, g[3].bias, g[4], relu=False))
        y = HF.conv2d(HF.gate_mul(input_features, a), self.weight, self.bias)
        return y, a


# ------------------------------------------------------------------------------------------------ dual attention decoder
class SEModule(nn.Module):
    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, channels // reduction, kernel_size=1, padding=0)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, padding=0)
        self.sigmoid = nn.Sigmoid()
        _init_conv_bn(self, (nn.Conv2d,))

    def forward(self, x):
        zero_s = torch.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
        return HF.dual_att_tail(x, zero_s, self.fc1, self.fc2)


class SpatialAttentionBlock(nn.Module):
    def __init__(self, in_features, attn_features, up_factor, normalize_attn=False):
        super().__init__()
        if normalize_attn:
            raise NotImplementedError("normalize_attn=True is never used by SAUNet")
        self.up_factor = up_factor
        self.normalize_attn = normalize_attn
        self.down = nn.Conv2d(in_features, attn_features, kernel_size=1, padding=0, bias=False)
        self.phi = nn.Conv2d(attn_features, 1, kernel_size=1, padding=0, bias=True)
        self.relu = nn.ReLU(inplace=True)
        self.bn = nn.BatchNorm2d(attn_features)
        _init_conv_bn(self, (nn.Conv2d,))

    def forward(self, x):
        c = HF.conv_bn_act(x, self.down.weight, None, self.bn, relu=True)
        return HF.sigmoid(HF.conv2d(c, self.phi.weight, self.phi.bias))


class _MRF(nn.Module):
    def __init__(self, inchannels):
        super().__init__()
        self.up = nn.Sequential(nn.ConvTranspose2d(inchannels[0], inchannels[0], kernel_size=4, stride=2, padding=1),
                                nn.BatchNorm2d(inchannels[0]), nn.ReLU(inplace=True))
        _init_conv_bn(self)

    def forward(self, channels, cat_buf=None):
        """cat_buf: optional preallocated concat buffer [N, C_skip + C_up, H, W] whose first C_skip channels ARE channels[1] (its
        producer wrote them there); the up-sampled branch is then written next to them and nothing is copied."""
        if len(channels) == 1:
            return channels[0]
        if cat_buf is not None:
            cs = channels[1].shape[1]
            up = HF.conv_bn_act(channels[0], self.up[0].weight, self.up[0].bias, self.up[1], relu=True, transposed=True, out=cat_buf[:, cs:])
            return HF.cat_alias(cat_buf, [channels[1], up])
        up = HF.conv_bn_act(channels[0], self.up[0].weight, self.up[0].bias, self.up[1], relu=True, transposed=True)
        return HF.cat([channels[1], up])


class DualAttBlock(nn.Module):
    def __init__(self, inchannels=[128, 256], outchannels=256):
        super().__init__()
        self.mrf = _MRF(inchannels)
        self.spatialAttn = SpatialAttentionBlock(outchannels, int(outchannels / 4), 2)
        self.channelAttn = SEModule(outchannels, 16)
        self.c3x3rb = ConvBNReLU(sum(inchannels), outchannels, 3, 1, 1)
        _init_conv_bn(self)

    def forward(self, x, cat_buf=None):
        cat = self.mrf(x, cat_buf)
        # channel attention squeeze: the global average pool of F is accumulated while F is written (BN-apply pass of c3x3rb)
        pooled = torch.empty(cat.shape[0], self.c3x3rb[0].out_channels, dtype=torch.float32, device=cat.device) if cat.is_cuda else None
        fused = self.c3x3rb(cat, pool=pooled)
        spatial = self.spatialAttn(fused)
        out = HF.dual_att_tail(fused, spatial, self.channelAttn.fc1, self.channelAttn.fc2, pooled)  # (1 + S) * SE(F)
        return out, spatial


class DecoderBlock(nn.Module):
    """models/models.py:203-237.  is_deconv=True (what SAUNet builds, dec1): conv3x3_bn_relu -> ConvTranspose2d(4, 2, 1) -> BN -> ReLU.
    is_deconv=False: nn.Upsample(x2, bilinear, align_corners=True) -> conv3x3_bn_relu -> conv3x3_bn_relu; the parameter-free Upsample keeps
    child index 0, so the state-dict keys are the reference's (block.1.0.weight ... block.2.1.running_var)."""

    def __init__(self, in_channels, middle_channels, out_channels, is_deconv=True):
        super().__init__()
        self.in_channels = in_channels
        self.is_deconv = bool(is_deconv)
        if is_deconv:
            self.block = nn.Sequential(conv3x3_bn_relu(in_channels, middle_channels),
                                       nn.ConvTranspose2d(middle_channels, out_channels, kernel_size=4, stride=2, padding=1),
                                       nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))
        else:
            self.block = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                                       conv3x3_bn_relu(in_channels, middle_channels), conv3x3_bn_relu(middle_channels, out_channels))
        _init_conv_bn(self, (nn.Conv2d,))

    def forward(self, x, out=None):
        b = self.block
        if self.is_deconv:
            return HF.conv_bn_act(b[0](x), b[1].weight, b[1].bias, b[2], relu=True, transposed=True, out=out)
        return b[2](b[1](HF.interpolate_bilinear(x, scale_factor=2)), out=out)


# ------------------------------------------------------------------------------------------------ selectable global pooling
def pooling_factor(pool_type="avg"):
    return 2 if pool_type == "avgmaxc" else 1


def adaptive_avgmax_pool2d(x, pool_type="avg", padding=0, count_include_pad=False):
    """models/adaptive_avgmax_pool.py:19-40 (global pooling: the kernel is the whole map, so padding must be 0)."""
    if padding != 0:
        raise NotImplementedError("global pooling over the whole map: padding=0")
    return HF.adaptive_avgmax_pool2d(x, pool_type)


class AdaptiveAvgMaxPool2d(nn.Module):
    """models/adaptive_avgmax_pool.py:43-74 with output_size=1."""

    def __init__(self, output_size=1, pool_type="avg"):
        super().__init__()
        if output_size not in (1, (1, 1)):
            raise NotImplementedError("AdaptiveAvgMaxPool2d: output_size=1 (global pooling)")
        self.output_size, self.pool_type = output_size, pool_type
        if pool_type not in HF.POOL_MODES:
            print("Invalid pool type %s specified. Defaulting to average pooling." % pool_type)

    def forward(self, x):
        return HF.adaptive_avgmax_pool2d(x, self.pool_type if self.pool_type in HF.POOL_MODES else "avg")

    def factor(self):
        return pooling_factor(self.pool_type)

    def __repr__(self):
        return self.__class__.__name__ + " (output_size=" + str(self.output_size) + ", pool_type=" + self.pool_type + ")"


# ------------------------------------------------------------------------------------------------ DenseNet-121 encoder
class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin); self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth); self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)


class _DenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, cin, growth=32, bn_size=4):
        super().__init__()
        for i in range(num_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(cin + i * growth, growth, bn_size)

    def forward(self, x, with_stats=False):
        buf, stats = HF.dense_block(x, list(self.values()), self.training)
        return (buf, stats) if with_stats else buf

    def total_channels(self):
        """channels of the block's concat buffer (input + growth * layers): what its producer may reserve (HF.reserve_dense_input)"""
        first = next(iter(self.values()))
        return first.norm1.num_features + sum(l.conv2.out_channels for l in self.values())


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(OrderedDict([("norm", nn.BatchNorm2d(cin)), ("relu", nn.ReLU(inplace=True)),
                                      ("conv", nn.Conv2d(cin, cout, 1, bias=False)), ("pool", nn.AvgPool2d(2, 2))]))

    def forward(self, x, stats=None, reserve=0):
        if stats is None and self.training:
            stats = HF.bn_stats(x)
        return HF.transition(x, stats, self, self.training, reserve)


class DenseNet121(nn.Module):
    """torchvision.models.densenet121 layout (features.* / classifier keys); only `features` is on the path."""

    def __init__(self, growth=32, blocks=(6, 12, 24, 16), init_features=64, bn_size=4, num_classes=1000):
        super().__init__()
        feats = OrderedDict([("conv0", nn.Conv2d(3, init_features, 7, stride=2, padding=3, bias=False)),
                             ("norm0", nn.BatchNorm2d(init_features)), ("relu0", nn.ReLU(inplace=True)),
                             ("pool0", nn.MaxPool2d(3, stride=2, padding=1))])
        c = init_features
        for b, n in enumerate(blocks):
            feats["denseblock%d" % (b + 1)] = _DenseBlock(n, c, growth, bn_size)
            c += n * growth
            if b != len(blocks) - 1:
                feats["transition%d" % (b + 1)] = _Transition(c, c // 2)
                c //= 2
        feats["norm5"] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(feats)
        self.classifier = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1); nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)


def densenet121(pretrained=False, **kw):
    """ImageNet weights cannot be downloaded here; load them with load_state_dict if you have them."""
    return DenseNet121(**kw)


class _Stem(nn.Sequential):
    """conv0 (7x7 s2, 3->64) + norm0, no ReLU / pool (models/models.py:304-305).  The 3-channel image is padded to 8
    channels (16-byte chunks), lowered by im2col to a [P, 7*7*8] matrix and multiplied on the 1x1 MFMA kernels
    (forward and weight gradient; the image needs no input gradient)."""

    def forward(self, x, reserve=0):
        """reserve: total channels of the dense block that consumes the result -- the output is then written as the first channel slice of
        that block's concat buffer (HF.reserve_dense_input) instead of being copied into it"""
        conv, bn = self[0], self[1]
        co, ci, kh, kw = conv.weight.shape
        w8 = torch.nn.functional.pad(conv.weight, (0, 0, 0, 0, 0, 8 - ci))                 # [64, 8, 7, 7]
        wk = w8.permute(0, 2, 3, 1).reshape(co, kh * kw * 8, 1, 1)                          # K order (kh, kw, c)
        cols = HF.im2col(x, kh, kw, conv.stride[0], conv.padding[0])
        out = None
        if reserve and reserve > co and x.is_cuda:
            out = HF.reserve_dense_input(cols.shape[0], co, cols.shape[2], cols.shape[3], reserve, cols.dtype, cols.device)
        return HF.conv_bn_act(cols, wk, None, bn, relu=False, out=out)


class _Tail(nn.Sequential):
    """denseblock4 + norm5 (no ReLU), models/models.py:312-313."""

    def forward(self, x, out=None):
        buf, stats = self[0](x, with_stats=True)
        return HF.batch_norm_act(buf, self[1], relu=False, stats=stats if self.training else None, out=out)


# ------------------------------------------------------------------------------------------------ SAUNet
class SAUNet(nn.Module):
    def __init__(self, num_classes=4, num_filters=32, pretrained=True, is_deconv=True, compute_dtype=None):
        super().__init__()
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype or _COMPUTE_DTYPE
        self.pool = nn.MaxPool2d(2, 2)
        self.encoder = densenet121(pretrained=pretrained)
        self.relu = nn.ReLU(inplace=True)
        self.sigmoid = nn.Sigmoid()
        # shape stream
        self.c3 = nn.Conv2d(256, 1, kernel_size=1)
        self.c4 = nn.Conv2d(512, 1, kernel_size=1)
        self.c5 = nn.Conv2d(1024, 1, kernel_size=1)
        self.d0 = nn.Conv2d(128, 64, kernel_size=1)
        self.res1 = BasicBlock(64, 64)
        self.d1 = nn.Conv2d(64, 32, kernel_size=1)
        self.res2 = BasicBlock(32, 32)
        self.d2 = nn.Conv2d(32, 16, kernel_size=1)
        self.res3 = BasicBlock(16, 16)
        self.d3 = nn.Conv2d(16, 8, kernel_size=1)
        self.fuse = nn.Conv2d(8, 1, kernel_size=1, padding=0, bias=False)
        self.cw = nn.Conv2d(2, 1, kernel_size=1, padding=0, bias=False)
        self.gate1 = GatedSpatialConv2d(32, 32)
        self.gate2 = GatedSpatialConv2d(16, 16)
        self.gate3 = GatedSpatialConv2d(8, 8)
        self.expand = ConvBNReLU(1, num_filters, kernel_size=1, padding=0)
        # encoder aliases (the same module objects, registered twice like the reference -> same state-dict keys)
        f = self.encoder.features
        self.conv1 = _Stem(f.conv0, f.norm0)
        self.conv2 = f.denseblock1
        self.conv2t = f.transition1
        self.conv3 = f.denseblock2
        self.conv3t = f.transition2
        self.conv4 = f.denseblock3
        self.conv4t = f.transition3
        self.conv5 = _Tail(f.denseblock4, f.norm5)
        # decoder
        self.center = conv3x3_bn_relu(1024, num_filters * 8 * 2)
        self.dec5 = DualAttBlock(inchannels=[512, 1024], outchannels=512)
        self.dec4 = DualAttBlock(inchannels=[512, 512], outchannels=256)
        self.dec3 = DualAttBlock(inchannels=[256, 256], outchannels=128)
        self.dec2 = DualAttBlock(inchannels=[128, 128], outchannels=64)
        self.dec1 = DecoderBlock(64, 48, num_filters, is_deconv)
        self.dec0 = conv3x3_bn_relu(num_filters * 2, num_filters)
        self.final = nn.Conv2d(num_filters, self.num_classes, kernel_size=1)

    def train(self, mode=True):
        if mode != self.training:
            HF._EVAL_BN.clear()       # eval-mode BatchNorm coefficients are cached per module: drop them on every mode switch
        return super().train(mode)

    def _prep_input(self, x):
        """float image [B,3,H,W] (any layout) -> NHWC, 8 channels (5 zero), compute dtype."""
        n, c, h, w = x.shape
        if c != 3:
            raise RuntimeError("SAUNet expects a 3-channel image")
        if h % 32 or w % 32:
            raise RuntimeError("SAUNet input H, W must be multiples of 32")
        xs = HF.nhwc(x.detach().to(torch.float32))
        x8 = HF.new_act(n, 8, h, w, self.compute_dtype, x.device, zero=True)
        HF.copy_channels(xs, x8[:, :3])
        return x8

    def _bump_counters(self):
        """all 150 BatchNorm `num_batches_tracked` counters live in one int64 tensor: one add per step"""
        bns = [m for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.num_batches_tracked is not None]
        flat = getattr(self, "_nbt_flat", None)
        ok = flat is not None and flat.device == bns[0].num_batches_tracked.device and all(
            b.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, b in enumerate(bns))
        if not ok:
            flat = torch.stack([b.num_batches_tracked.detach().reshape(()) for b in bns]).contiguous()
            for i, b in enumerate(bns):
                b.num_batches_tracked = flat[i]
                b._nbt_fused = True
            object.__setattr__(self, "_nbt_flat", flat)
        flat.add_(1)

    def forward(self, x, return_att=False):
        HF.begin_step()
        if self.training:
            self._bump_counters()
        size = x.shape[2:]
        up = HF.interpolate_bilinear
        n, dev, dt = x.shape[0], x.device, self.compute_dtype
        h16, w16 = size[0] // 16, size[1] // 16
        # decoder concatenations [skip | up-sampled] are preallocated; their producers write straight into the channel slices
        cat5 = HF.new_act(n, 1024 + 512, h16, w16, dt, dev)
        cat4 = HF.new_act(n, 512 + 512, 2 * h16, 2 * w16, dt, dev)
        cat3 = HF.new_act(n, 256 + 256, 4 * h16, 4 * w16, dt, dev)
        cat2 = HF.new_act(n, 128 + 128, 8 * h16, 8 * w16, dt, dev)
        nf = self.final.in_channels
        cat0 = HF.new_act(n, 2 * nf, size[0], size[1], dt, dev)
        grad = torch.is_grad_enabled() or self.training            # (the inference block builds its own buffer)
        conv1 = self.conv1(self._prep_input(x), reserve=self.conv2.total_channels() if grad else 0)
        buf, st = self.conv2(conv1, with_stats=True); conv2 = self.conv2t(buf, st, reserve=self.conv3.total_channels() if grad else 0)
        buf, st = self.conv3(conv2, with_stats=True); conv3 = self.conv3t(buf, st, reserve=self.conv4.total_channels() if grad else 0)
        buf, st = self.conv4(conv3, with_stats=True); conv4 = self.conv4t(buf, st, reserve=self.conv5[0].total_channels() if grad else 0)
        conv5 = self.conv5(conv4, out=cat5[:, :1024])

        def conv(m, t):
            return HF.conv2d(t, m.weight, m.bias)

        # (every residual block of the shape stream feeds exactly one 1x1 convolution: block + convolution are one autograd node, see
        # functional._BasicBlockConv)
        ss = HF.basic_block_conv1x1(up(conv(self.d0, conv2), size), self.res1, self.d1)
        c3 = up(conv(self.c3, conv3), size)
        ss, g1 = self.gate1(ss, c3)
        ss = HF.basic_block_conv1x1(ss, self.res2, self.d2)
        c4 = up(conv(self.c4, conv4), size)
        ss, g2 = self.gate2(ss, c4)
        ss = HF.basic_block_conv1x1(ss, self.res3, self.d3)
        c5 = up(conv(self.c5, conv5), size)
        ss, g3 = self.gate3(ss, c5)
        # the one/two-channel edge head stays in float32 whatever the storage dtype: a bf16 sigmoid saturates to
        # exactly 1.0 and BCE's log(1 - e) would hit its -100 clamp (the reference is float32 throughout)
        ss = HF.cast(up(conv(self.fuse, ss), size), torch.float32)
        edge_out = HF.sigmoid(ss)

        canny = HF.canny(x, 10, 100, dtype=torch.float32)                 # on device, no host round trip
        acts = HF.sigmoid(conv(self.cw, HF.cat([edge_out, canny])))
        edge = self.expand(acts, out_dtype=self.compute_dtype, out=cat0[:, nf:])

        conv2u = up(conv2, scale_factor=2, out=cat2[:, :128])
        conv3u = up(conv3, scale_factor=2, out=cat3[:, :256])
        conv4u = up(conv4, scale_factor=2, out=cat4[:, :512])
        center = self.center(HF.max_pool2x2(conv5))
        dec5, att5 = self.dec5([center, conv5], cat5)
        dec4, att4 = self.dec4([dec5, conv4u], cat4)
        dec3, att3 = self.dec3([dec4, conv3u], cat3)
        dec2, att2 = self.dec2([dec3, conv2u], cat2)
        dec1 = self.dec1(dec2, out=cat0[:, :nf])
        dec0 = self.dec0(HF.cat_alias(cat0, [dec1, edge]))
        x_out = conv(self.final, dec0)
        if return_att:
            with torch.no_grad():
                maps = [up(att2, scale_factor=2), up(att3, scale_factor=4), up(att4, scale_factor=8), up(att5, scale_factor=16)]
            return x_out, edge_out, maps + [g1, g2, g3]
        return x_out, edge_out


# ------------------------------------------------------------------------------------------------ loss / task wrapper
class DualLoss(nn.Module):
    def __init__(self, num_classes=4, lmbda=10, epsilon=10e-6, mode="train"):
        super().__init__()
        if num_classes != 4:
            raise NotImplementedError("the fused loss kernel is specialised for the 4 ACDC classes")
        self.epsilon, self.lmbda, self.channels = epsilon, lmbda, num_classes
        self.epoch, self.alpha = 1, 1.0
        self.last_metrics = None

    def forward(self, pred, target, epoch=0):
        seg, edge_in = pred
        seg_t, edge_t = target
        if seg_t.dim() == 2:
            seg_t = seg_t.unsqueeze(0)
        loss, metrics = HF.dual_loss(seg, edge_in, seg_t, edge_t)
        self.last_metrics = metrics
        return loss


class SegmentationModuleBase(nn.Module):
    """models/models.py:20-78.  The training branch takes its metrics from the fused loss kernel (one pass over the logits: `_fused_metrics`);
    `pixel_acc` / `jaccard` with the reference's own signatures are thin device implementations for callers that hold a prediction tensor."""

    @staticmethod
    def _fused_metrics(metrics, num_class):
        return metrics[0], [metrics[i] for i in range(1, num_class)]

    def pixel_acc(self, pred, label, num_class):
        """models/models.py:51-74: pred [N, C, H, W] class scores (the reference passes round(softmax).long()), label [N, H, W];
        -> (acc over the labelled pixels, [Jaccard of class 1 .. num_class - 1]) as 0-d device tensors."""
        m = HF.pixel_metrics(pred, label, int(num_class))
        return m[0], [m[i] for i in range(1, int(num_class))]

    def jaccard(self, pred, label):
        """models/models.py:76-78: |pred & label| / (|pred| + |label| - |pred & label|) over binary masks."""
        return HF.binary_jaccard(pred, label)

    def intersectionAndUnion(self, imPred, imLab, numClass):
        """models/models.py:24-49 (host-side numpy, as in the reference): mean Jaccard of classes 1 and 2 of two label maps."""
        import numpy as np
        imPred = np.asarray(imPred.cpu() if torch.is_tensor(imPred) else imPred).copy()
        imLab = np.asarray(imLab.cpu() if torch.is_tensor(imLab) else imLab).copy()
        imPred += 1
        imLab += 1
        imPred = imPred * (imLab > 0)
        inter = imPred * (imPred == imLab)
        ai, _ = np.histogram(inter, bins=numClass, range=(1, numClass))
        ap, _ = np.histogram(imPred, bins=numClass, range=(1, numClass))
        al, _ = np.histogram(imLab, bins=numClass, range=(1, numClass))
        with np.errstate(divide="ignore", invalid="ignore"):
            j = ai / (ap + al - ai)
        j = (j[1] + j[2]) / 2
        return j if j <= 1 else 0


class SegmentationModule(SegmentationModuleBase):
    def __init__(self, crit, unet, num_class):
        super().__init__()
        self.crit, self.unet, self.num_class = crit, unet, num_class

    def forward(self, feed_dict, epoch, *, segSize=None, return_att=False):
        if segSize is None:  # training
            p = self.unet(feed_dict["image"])
            loss = self.crit(p, feed_dict["mask"], epoch=epoch)
            return loss, self._fused_metrics(self.crit.last_metrics, self.num_class)
        if segSize is True:  # test
            p, e, maps = self.unet(feed_dict["image"], return_att=True)
            return HF.softmax_argmax(p, want_label=False)[0], maps
        out = self.unet(feed_dict["image"], return_att=return_att)
        seg_t, edge_t = feed_dict["mask"]
        loss = self.crit((out[0], out[1]), (seg_t.long().unsqueeze(0), edge_t.unsqueeze(0)))
        return HF.softmax_argmax(out[0], want_label=False)[0], loss


class ModelBuilder:
    def build_unet(self, num_class=1, arch="saunet", weights=""):
        if arch.lower() != "saunet":
            raise Exception("Architecture undefined!")
        unet = SAUNet(num_classes=num_class)
        if len(weights) > 0:
            unet.load_state_dict(torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return unet
