"""Multi-exit ResNet-18 family behind the reference's API (SA/models/resnet18/resnet18.py).

Same class names, constructor arguments, attribute names and ``state_dict`` keys as the
reference (``ResNet`` :88-180, ``ResNet18EarlyExit`` :182-186, ``ResNet18Base`` :189-204,
``MCDropout`` :207-210, ``ResNet18MC`` :212-258, ``ResNet18MCEarlyExit`` :260-346), and the same
construction order, so a given torch seed yields the reference's initial weights.  These
modules are parameter containers and graph descriptions: ``model(x)`` compiles the module
tree into the HIP engine (bayesnn_fpga_amd/engine.py) and runs ONE stochastic pass there,
returning ``[logits per exit]`` like the reference's forward.  There is no CPU forward.
"""
import math

import torch
from torch import nn

from ...utils import Masksembles1D, Masksembles2D
from .._engine_mixin import EngineModelMixin

_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))           # (planes, first-block stride)
_EXIT_CONVS = {1: ((64, 128), (128, 256), (256, 512)), 2: ((128, 256), (256, 512)), 3: ((256, 512),)}


class MCDropout(nn.Dropout):
    """Dropout that stays on at inference (reference :207-210).  Executes as a Philox-masked
    epilogue / prologue inside the HIP kernels; ``p`` is all the engine reads from it."""

    def forward(self, x):
        raise RuntimeError("MCDropout executes inside the HIP engine; bayesnn_fpga_amd has no CPU path")


def _no_cpu(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} executes inside the HIP engine; bayesnn_fpga_amd has no CPU path")


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.planes = planes
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    forward = _no_cpu


class ResNet(EngineModelMixin, nn.Module):
    family = "resnet"
    multi_exit = True

    def __init__(self, block=BasicBlock, num_blocks=(2, 2, 2, 2), num_classes=100):
        super().__init__()
        if block is not BasicBlock:
            raise NotImplementedError("only BasicBlock networks are on the accelerated path")
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        for i, ((planes, stride), nb) in enumerate(zip(_STAGES, num_blocks), 1):
            setattr(self, f"layer{i}", self._make_layer(block, planes, nb, stride))
        self.linear = nn.Linear(512 * block.expansion, num_classes)
        for e, convs in _EXIT_CONVS.items():
            for j, (cin, cout) in enumerate(convs, 1):
                setattr(self, f"ex{e}conv{j}", nn.Conv2d(cin, cout, 3, 2, 1, bias=False))
            for j, (_, cout) in enumerate(convs, 1):
                setattr(self, f"ex{e}bn{j}", nn.BatchNorm2d(cout))
            setattr(self, f"ex{e}linear", nn.Linear(512, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._init_engine_state()

    def _make_layer(self, block, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)


class ResNet18EarlyExit(ResNet):
    def __init__(self, n_exits=4, out_dim=100, image_size=32, *args, **kwargs):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim, *args, **kwargs)
        self.n_exits = n_exits
        self.out_dim = out_dim


class ResNet18Base(ResNet):
    multi_exit = False

    def __init__(self, n_exits=1, out_dim=100, *args, **kwargs):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim, *args, **kwargs)
        self.n_exits = n_exits
        self.out_dim = out_dim


def _configure_mc(model, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale):
    model.n_exits, model.out_dim = n_exits, out_dim
    model.dropout_exit, model.dropout, model.dropout_p = dropout_exit, dropout, dropout_p
    model.mask_type, model.num_masks, model.mask_scale = mask_type, num_masks, mask_scale
    stages = [model.layer1, model.layer2, model.layer3, model.layer4]

    def site(channels):
        return MCDropout(dropout_p) if mask_type == "mc" else Masksembles2D(channels, num_masks, mask_scale)

    if dropout == "block":              # after layer1..3, never after layer4 (reference :225-232, :273-280)
        for i in range(3):
            stages[i] = nn.Sequential(stages[i], site(stages[i][-1].planes))
        model.layer1, model.layer2, model.layer3, model.layer4 = stages
    elif dropout == "layer":            # after every BasicBlock but the very last (:233-240, :281-288)
        for si, stage in enumerate(stages):
            for bi in range(len(stage)):
                if si == 3 and bi == len(stage) - 1:
                    continue
                if mask_type != "mc":
                    # the reference dereferences an unbound loop variable here
                    raise UnboundLocalError("dropout='layer' with mask_type='mask' is broken in the reference")
                stage[bi] = nn.Sequential(stage[bi], MCDropout(dropout_p))


def _exit_site(model, channels):
    if model.mask_type == "mc":
        return MCDropout(model.dropout_p)
    return Masksembles1D(channels, model.num_masks, model.mask_scale)


class ResNet18MC(ResNet):
    multi_exit = False

    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=1, out_dim=100, image_size=32,
                 mask_type="mc", num_masks=4, mask_scale=4.0, *args, **kwargs):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim, *args, **kwargs)
        _configure_mc(self, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale)
        if dropout_exit:
            self.exit_dropout = _exit_site(self, 512 * BasicBlock.expansion)


class ResNet18MCEarlyExit(ResNet):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=4, out_dim=100, image_size=32,
                 mask_type="mc", num_masks=4, mask_scale=4.0, *args, **kwargs):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim, *args, **kwargs)
        _configure_mc(self, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale)
        if dropout_exit:
            self.exit1_dropout = _exit_site(self, 512)
            self.exit2_dropout = _exit_site(self, 512)
            self.exit3_dropout = _exit_site(self, 512)
            self.exit_dropout = _exit_site(self, 512 * BasicBlock.expansion)
