"""Import the real reference (/root/reference) on CPU -- CONTAINER-ONLY test infrastructure.

Used by make_golden.py (fixture generation) and tests/test_oracle_vs_reference.py
(skipped when /root/reference is absent, i.e. on the GPU box).  Nothing from the
reference is copied: it is imported where it lies, with stand-ins for the
third-party modules that are not installed:

* ``torchvision``  -> oracle/ref_shims/torchvision (DenseNet-121 restatement)
* ``cv2``          -> oracle/ref_shims/cv2.py     (Canny restatement)
* ``.cuda()``      -> identity (the reference hard-codes .cuda(): loss.py:130,132,153-156,
                      models/models.py:92,363)
* nibabel / imageio / skimage / scipy.misc -> empty modules (data loader imports only)
"""
import os
import sys
import types

REF = "/root/reference"
_loaded = {}


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def load():
    """Returns a namespace with the reference's SAUNet, SegmentationModule, DualLoss, ... ."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference not present (GPU box?)")
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    import torch

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(here, "ref_shims"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    # order matters: shims must win over anything else; REF last-inserted would shadow `oracle`? no:
    sys.path.remove(os.path.join(here, "ref_shims"))
    sys.path.insert(0, os.path.join(here, "ref_shims"))
    for name in ("nibabel", "imageio", "skimage", "skimage.transform", "scipy.misc"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "skimage.transform":
                m.resize = None
            sys.modules[name] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import models.models as ref_models  # noqa
        import models.attention_blocks as ref_att
        import models.GSConv as ref_gsc
        import models.resnet as ref_resnet
        import loss as ref_loss
        import utils as ref_utils
        import radam as ref_radam
    ns = types.SimpleNamespace(
        models=ref_models, att=ref_att, gsc=ref_gsc, resnet=ref_resnet, loss=ref_loss,
        utils=ref_utils, radam=ref_radam,
        SAUNet=ref_models.SAUNet, SegmentationModule=ref_models.SegmentationModule,
        DecoderBlock=ref_models.DecoderBlock, conv3x3_bn_relu=ref_models.conv3x3_bn_relu,
        DualAttBlock=ref_att.DualAttBlock, SEModule=ref_att.SEModule,
        SpatialAttentionBlock=ref_att.SpatialAttentionBlock, MRF=ref_att._MRF,
        GatedSpatialConv2d=ref_gsc.GatedSpatialConv2d, BasicBlock=ref_resnet.BasicBlock,
        DualLoss=ref_loss.DualLoss, dice_loss=ref_loss.dice_loss,
        intersectionAndUnion=ref_utils.intersectionAndUnion, RAdam=ref_radam.RAdam,
    )
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import data.ac17_dataloader as ref_data
        ns.AC17_2DLoad = ref_data.AC17_2DLoad
    except Exception as e:  # pragma: no cover - data module needs more third-party stubs
        ns.AC17_2DLoad = None
        ns.data_import_error = repr(e)
    _loaded["ns"] = ns
    return ns
