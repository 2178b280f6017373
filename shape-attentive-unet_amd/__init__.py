"""saunet_amd -- MI355X-native (gfx950) SAUNet training/inference hot path.

Import name is ``saunet_amd`` (the directory is ``shape-attentive-unet_amd/``; the repo-root
``saunet_amd.py`` shim maps one onto the other).  Everything numerical runs in
``libsaunet_hip.so`` (hand-written HIP, see csrc/); there is no CPU or PyTorch fallback.
"""
from . import lib  # noqa: F401
from . import functional  # noqa: F401
from .modules import (  # noqa: F401
    SAUNet, SegmentationModule, SegmentationModuleBase, ModelBuilder, DualLoss, DualAttBlock, SEModule,
    SpatialAttentionBlock, GatedSpatialConv2d, BasicBlock, DecoderBlock, conv3x3_bn_relu, ConvBNReLU, Norm2d,
    SynchronizedBatchNorm2d, DenseNet121, densenet121, set_compute_dtype, get_compute_dtype, AdaptiveAvgMaxPool2d,
    adaptive_avgmax_pool2d, pooling_factor)
from . import optim  # noqa: F401
from . import dp  # noqa: F401
from . import graph  # noqa: F401
from . import postprocess  # noqa: F401
from . import augment  # noqa: F401
from . import nifti  # noqa: F401
from . import acdc  # noqa: F401
from . import dice  # noqa: F401

__version__ = "0.1.0"
