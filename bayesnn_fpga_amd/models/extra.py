"""Build-defined models for the BASELINE configs that have NO PyTorch counterpart in the reference
(SURVEY.md §8.0): VGG-11-BN with dropout before its dense layers (config 2) and a multi-exit ResNet-50
(config 5).  Architectures are taken from the reference's Keras definitions / its ``Bottleneck`` block:

* ``VGG11MC``             — Hardware_Artifact/bayes_hw/models/models.py:211-287 (64 filters, dense 512/512/C; the
  ``num_bayes_layer`` counter rule over its 7 candidate sites; 3 = dropout before each dense layer).
* ``ResNet50MCEarlyExit`` — SA/models/resnet18/resnet18.py:51-85 (Bottleneck), :88-180 with [3,4,6,3]; the first conv
  of every exit head takes the stage's real width (the reference's 64/128/256-channel heads make the
  Bottleneck network fail in ``ex1conv1``).
Parity for these is pinned against the fp32 CPU restatement in ``oracle/extra_models.py`` only.  Dense layers run
as 1x1 convolutions on [N,1,1,C] tensors (same MFMA kernel, same fused dropout epilogue).
LeNet-5 (config 1) is "CPU plumbing, no GPU" in BASELINE and is not on the accelerated path.
"""
import math

from torch import nn

from ._engine_mixin import EngineModelMixin
from .resnet18.resnet18 import MCDropout, _no_cpu

VGG11_CFG = (64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M')


def _he_init(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class VGG11MC(EngineModelMixin, nn.Module):
    family = "vgg11"
    multi_exit = False

    def __init__(self, num_bayes_layer=3, dropout_p=0.25, out_dim=10, dense=(512, 512)):
        super().__init__()
        self.n_exits, self.out_dim, self.dropout_p, self.num_bayes_layer = 1, out_dim, dropout_p, num_bayes_layer
        first_site = 7 - num_bayes_layer
        feats, cin, loc = [], 3, 0
        for v in VGG11_CFG:
            if v == 'M':
                feats.append(nn.MaxPool2d(2, 2))
                if loc < 4:
                    if loc >= first_site:
                        feats.append(MCDropout(dropout_p))
                    loc += 1
            else:
                feats += [nn.Conv2d(cin, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*feats)
        cls, cin = [], 512
        if 4 >= first_site:
            cls.append(MCDropout(dropout_p))
        for i, d in enumerate(dense):
            cls += [nn.Linear(cin, d), nn.ReLU(inplace=True)]
            if 5 + i >= first_site:
                cls.append(MCDropout(dropout_p))
            cin = d
        cls.append(nn.Linear(cin, out_dim))
        self.classifier = nn.Sequential(*cls)
        _he_init(self)
        self._init_engine_state()

    def build_graph(self, g):
        x = g.tensor(32, 32, 3)
        mods, i, first = list(self.features), 0, True
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                x = g.conv(x, m, mods[i + 1], relu=True, stem=first)
                first = False
                i += 3
            elif isinstance(m, nn.MaxPool2d):
                x = g.maxpool(x)
                i += 1
            elif isinstance(m, nn.Dropout):
                x = g.mask(x, g.site(m))
                i += 1
            else:
                raise TypeError(type(m).__name__)
        cls, i = list(self.classifier), 0
        pending = None                        # a site that precedes the next dense layer
        while i < len(cls):
            m = cls[i]
            if isinstance(m, nn.Dropout):
                pending = m
                i += 1
            elif isinstance(m, nn.Linear) and i + 1 < len(cls):       # hidden dense + ReLU (+ dropout)
                if pending is not None:
                    x = g.mask(x, g.site(pending))
                    pending = None
                nxt = cls[i + 2] if i + 2 < len(cls) and isinstance(cls[i + 2], nn.Dropout) else None
                x = g.dense(x, m, relu=True, site=g.site(nxt))
                i += 3 if nxt is not None else 2
            else:                                                       # final classifier
                g.head(x, m, 0, site=g.site(pending))
                i += 1


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

    forward = _no_cpu


class ResNet50MCEarlyExit(EngineModelMixin, nn.Module):
    family = "resnet50"
    multi_exit = True

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
        self._init_engine_state()

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def build_graph(self, g):
        from ..engine import _unwrap
        x = g.tensor(32, 32, 3)
        x = g.conv(x, self.conv1, self.bn1, relu=False, stem=True)
        for si in range(1, 5):
            stage, stage_site = _unwrap(getattr(self, f"layer{si}"))
            blocks = list(stage)
            for bi, blk in enumerate(blocks):
                site_mod = stage_site if bi == len(blocks) - 1 else None
                a = g.conv(x, blk.conv1, blk.bn1, relu=True)
                b = g.conv(a, blk.conv2, blk.bn2, relu=True)
                res = x if blk.downsample is None else g.conv(x, blk.downsample[0], blk.downsample[1], relu=False)
                x = g.conv(b, blk.conv3, blk.bn3, relu=True, residual=res, site=g.site(site_mod))
            if si < 4:
                y = x
                for j in range(1, 5 - si):
                    y = g.conv(y, getattr(self, f"ex{si}conv{j}"), getattr(self, f"ex{si}bn{j}"), relu=True)
                sm = getattr(self, f"exit{si}_dropout", None) if self.dropout_exit else None
                g.head(y, getattr(self, f"ex{si}linear"), si - 1, site=g.site(sm))
        sm = getattr(self, "exit_dropout", None) if self.dropout_exit else None
        g.head(x, self.linear, 3, site=g.site(sm))
