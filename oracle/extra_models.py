"""Build-defined torch restatements for the BASELINE configs that have NO PyTorch reference
(TEST ORACLE — see oracle/__init__.py): parity for these is pinned only against
this fp32 CPU restatement, not against the reference.

* ``LeNet5MC``   — config 1.  Architecture from the Keras model Hardware_Artifact/bayes_hw/models/models.py:34-73:
  conv5x5x20 same + ReLU, maxpool 2, conv5x5x20 same + ReLU, maxpool 7 (28x28 -> 2x2), flatten(80), Dense 100 +
  ReLU, [dropout], Dense 10.  One Bayesian layer = the last site of its counter rule.  CPU only (its 1- and
  20-channel 5x5 convs are outside the HIP kernels' range; BASELINE calls this config "plumbing, no GPU").
* ``VGG11MC``    — config 2.  Keras Hardware_Artifact/bayes_hw/models/models.py:211-287 (64-filter version of
  autobayes/models/VGG.py:12-53): 7 candidate sites (after each of the 4 first max-pools, after flatten, after
  dense0, after dense1); ``num_bayes_layer=3`` puts elementwise dropout before each of the 3 dense layers.
* ``ResNet50MCEarlyExit`` — config 5.  The reference's ResNet with ``block=Bottleneck, [3,4,6,3]``
  (SA/models/resnet18/resnet18.py:51-85, :88-180) fails in its exit heads (64-ch weights vs 256-ch input); here the
  first conv of each exit head takes the stage's real width (256/512/1024) and the final FC is 2048-wide.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import MCContext, MCDropout


def _he_init(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class _MCBase(nn.Module):
    def _finish(self):
        self.mc = MCContext()
        for m in self.modules():
            if isinstance(m, MCDropout):
                m.ctx = self.mc


class LeNet5MC(_MCBase):
    def __init__(self, dropout_p=0.2, out_dim=10):
        super().__init__()
        self.n_exits, self.out_dim, self.dropout_p = 1, out_dim, dropout_p
        self.conv1 = nn.Conv2d(1, 20, 5, padding=2)
        self.conv2 = nn.Conv2d(20, 20, 5, padding=2)
        self.fc1 = nn.Linear(80, 100)
        self.drop = MCDropout(dropout_p)
        self.fc2 = nn.Linear(100, out_dim)
        self._finish()

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 7)
        x = F.relu(self.fc1(x.flatten(1)))
        return [self.fc2(self.drop(x))]


VGG11_CFG = (64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M')


class VGG11MC(_MCBase):
    def __init__(self, num_bayes_layer=3, dropout_p=0.25, out_dim=10, dense=(512, 512)):
        super().__init__()
        self.n_exits, self.out_dim, self.dropout_p, self.num_bayes_layer = 1, out_dim, dropout_p, num_bayes_layer
        first_site = 7 - num_bayes_layer                # sites are locations first_site .. 6
        feats, cin, loc = [], 3, 0
        for v in VGG11_CFG:
            if v == 'M':
                feats.append(nn.MaxPool2d(2, 2))
                if loc < 4:                             # only the first four pools are candidate locations
                    if loc >= first_site:
                        feats.append(MCDropout(dropout_p))
                    loc += 1
            else:
                feats += [nn.Conv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*feats)
        cls, cin = [], 512
        if 4 >= first_site:
            cls.append(MCDropout(dropout_p))            # after flatten
        for i, d in enumerate(dense):
            cls += [nn.Linear(cin, d), nn.ReLU(inplace=True)]
            if 5 + i >= first_site:
                cls.append(MCDropout(dropout_p))
            cin = d
        cls.append(nn.Linear(cin, out_dim))
        self.classifier = nn.Sequential(*cls)
        _he_init(self)
        self._finish()

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        return [self.classifier(self.features(x).flatten(1))]


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.planes = planes
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        res = x if self.downsample is None else self.downsample(x)
        return F.relu(out + res)


class ResNet50MCEarlyExit(_MCBase):
    def __init__(self, dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10, num_blocks=(3, 4, 6, 3)):
        super().__init__()
        self.n_exits, self.out_dim = 4, out_dim
        self.dropout_exit, self.dropout, self.dropout_p, self.mask_type = dropout_exit, dropout, dropout_p, "mc"
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), 1):
            setattr(self, f"layer{i}", self._make_layer(planes, num_blocks[i - 1], stride))
        self.linear = nn.Linear(2048, out_dim)
        for e, chans in ((1, (256, 128, 256, 512)), (2, (512, 256, 512)), (3, (1024, 512))):
            for j, (a, b) in enumerate(zip(chans[:-1], chans[1:]), 1):
                setattr(self, f"ex{e}conv{j}", nn.Conv2d(a, b, 3, 2, 1, bias=False))
                setattr(self, f"ex{e}bn{j}", nn.BatchNorm2d(b))
            setattr(self, f"ex{e}linear", nn.Linear(512, out_dim))
        _he_init(self)
        if dropout == "block":
            for i in (1, 2, 3):
                setattr(self, f"layer{i}", nn.Sequential(getattr(self, f"layer{i}"), MCDropout(dropout_p)))
        elif dropout is not None:
            raise ValueError("only dropout in {None, 'block'} is defined for the ResNet-50 config")
        if dropout_exit:
            for name in ("exit1_dropout", "exit2_dropout", "exit3_dropout", "exit_dropout"):
                setattr(self, name, MCDropout(dropout_p))
        self._finish()

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _head(self, e, out, n_conv):
        for j in range(1, n_conv + 1):
            out = getattr(self, f"ex{e}bn{j}")(getattr(self, f"ex{e}conv{j}")(F.relu(out)))
        out = F.avg_pool2d(F.relu(out), 4).flatten(1)
        if self.dropout_exit:
            out = getattr(self, f"exit{e}_dropout")(out)
        return getattr(self, f"ex{e}linear")(out)

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        out = self.layer1(self.bn1(self.conv1(x)))
        o1 = self._head(1, out, 3)
        out = self.layer2(out)
        o2 = self._head(2, out, 2)
        out = self.layer3(out)
        o3 = self._head(3, out, 1)
        out = F.avg_pool2d(F.relu(self.layer4(out)), 4).flatten(1)
        if self.dropout_exit:
            out = self.exit_dropout(out)
        return [o1, o2, o3, self.linear(out)]
