"""DenseNet-121 feature extractor, restated from the public topology.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference calls the third-party ``torchvision.models.densenet121``
(/root/reference/models/models.py:271) and slices its ``features`` into
conv1..conv5 (/root/reference/models/models.py:304-313).  torchvision is not
vendored in the reference and not installed here (requirements.txt:2 gives no
version pin), so this file restates the published architecture
(Huang et al. 2017; growth 32, blocks (6,12,24,16), 64 init features,
bn_size 4, no dropout) with torchvision's state-dict key names:

    features.conv0 / norm0 / relu0 / pool0
    features.denseblock{b}.denselayer{i}.{norm1,relu1,conv1,norm2,relu2,conv2}
    features.transition{b}.{norm,relu,conv,pool}
    features.norm5
    classifier

Only ``torch.nn`` CPU primitives are used, so the arithmetic is the installed
torch's.  PARITY UNPINNED against real torchvision (none available offline).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

GROWTH = 32
BLOCKS = (6, 12, 24, 16)
INIT_FEATURES = 64
BN_SIZE = 4


class DenseLayer(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, BN_SIZE * GROWTH, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1) if isinstance(feats, (list, tuple)) else feats
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class DenseBlock(nn.ModuleDict):
    def __init__(self, nlayers, cin):
        super().__init__()
        for i in range(nlayers):
            self["denselayer%d" % (i + 1)] = DenseLayer(cin + i * GROWTH)

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(OrderedDict([
            ("norm", nn.BatchNorm2d(cin)),
            ("relu", nn.ReLU(inplace=True)),
            ("conv", nn.Conv2d(cin, cout, 1, bias=False)),
            ("pool", nn.AvgPool2d(2, 2)),
        ]))


class DenseNet121(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        feats = OrderedDict([
            ("conv0", nn.Conv2d(3, INIT_FEATURES, 7, stride=2, padding=3, bias=False)),
            ("norm0", nn.BatchNorm2d(INIT_FEATURES)),
            ("relu0", nn.ReLU(inplace=True)),
            ("pool0", nn.MaxPool2d(3, stride=2, padding=1)),
        ])
        c = INIT_FEATURES
        for b, n in enumerate(BLOCKS):
            feats["denseblock%d" % (b + 1)] = DenseBlock(n, c)
            c += n * GROWTH
            if b != len(BLOCKS) - 1:
                feats["transition%d" % (b + 1)] = Transition(c, c // 2)
                c //= 2
        feats["norm5"] = nn.BatchNorm2d(c)
        self.features = nn.Sequential(feats)
        self.classifier = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        f = F.relu(self.features(x), inplace=True)
        f = F.adaptive_avg_pool2d(f, (1, 1)).flatten(1)
        return self.classifier(f)


def densenet121(pretrained=False, **kw):
    """ImageNet weights cannot be fetched offline: always random init."""
    return DenseNet121(**kw)
