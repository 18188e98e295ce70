"""Multi-exit ResNet-18 family, restated (TEST ORACLE — see oracle/__init__.py).

Follows SA/models/resnet18/resnet18.py: BasicBlock :17-48, ResNet :88-180,
ResNet18EarlyExit :182-186, ResNet18Base :189-204, ResNet18MC :212-258,
ResNet18MCEarlyExit :260-346; factory SA/models/resnet18/resnet18_loader.py:4-16.
Module construction order and attribute names match the reference so that (a) the same
torch seed yields the same initial weights and (b) ``state_dict`` keys are identical
(SURVEY.md Appendix B).  Reference quirks kept: no ReLU after the stem (:303); block
dropout is never added after layer4 (:275-276); ``dropout="layer"`` with
``mask_type="mask"`` raises (the reference hits an undefined name at :288).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import MCContext, MCDropout, Masksembles1D, Masksembles2D


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.planes = planes
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out = out + residual
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, block=BasicBlock, num_blocks=(2, 2, 2, 2), num_classes=100):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.layer1 = self._make_layer(block, 64, num_blocks[0], stride=1)
        self.layer2 = self._make_layer(block, 128, num_blocks[1], stride=2)
        self.layer3 = self._make_layer(block, 256, num_blocks[2], stride=2)
        self.layer4 = self._make_layer(block, 512, num_blocks[3], stride=2)
        self.linear = nn.Linear(512 * block.expansion, num_classes)

        self.ex1conv1 = nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex1conv2 = nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex1conv3 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex1bn1 = nn.BatchNorm2d(128)
        self.ex1bn2 = nn.BatchNorm2d(256)
        self.ex1bn3 = nn.BatchNorm2d(512)
        self.ex1linear = nn.Linear(512, num_classes)

        self.ex2conv1 = nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex2conv2 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex2bn1 = nn.BatchNorm2d(256)
        self.ex2bn2 = nn.BatchNorm2d(512)
        self.ex2linear = nn.Linear(512, num_classes)

        self.ex3conv1 = nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1, bias=False)
        self.ex3bn1 = nn.BatchNorm2d(512)
        self.ex3linear = nn.Linear(512, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

        self.mc = MCContext()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # -- shared pieces --------------------------------------------------------------
    def _attach_ctx(self):
        for m in self.modules():
            if isinstance(m, MCDropout):
                m.ctx = self.mc

    def _exit1(self, out):
        out1 = self.ex1bn1(self.ex1conv1(F.relu(out)))
        out1 = self.ex1bn2(self.ex1conv2(F.relu(out1)))
        out1 = self.ex1bn3(self.ex1conv3(F.relu(out1)))
        out1 = F.avg_pool2d(F.relu(out1), 4)
        return out1.view(out1.size(0), -1)

    def _exit2(self, out):
        out2 = self.ex2bn1(self.ex2conv1(F.relu(out)))
        out2 = self.ex2bn2(self.ex2conv2(F.relu(out2)))
        out2 = F.avg_pool2d(F.relu(out2), 4)
        return out2.view(out2.size(0), -1)

    def _exit3(self, out):
        out3 = self.ex3bn1(self.ex3conv1(F.relu(out)))
        out3 = F.avg_pool2d(F.relu(out3), 4)
        return out3.view(out3.size(0), -1)

    def _final(self, out):
        out = F.avg_pool2d(F.relu(out), 4)
        return out.view(out.size(0), -1)

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        out = self.bn1(self.conv1(x))
        out = self.layer1(out)
        out1 = self.ex1linear(self._exit1(out))
        out = self.layer2(out)
        out2 = self.ex2linear(self._exit2(out))
        out = self.layer3(out)
        out3 = self.ex3linear(self._exit3(out))
        out = self.layer4(out)
        out = self.linear(self._final(out))
        return [out1, out2, out3, out]


class ResNet18EarlyExit(ResNet):
    def __init__(self, n_exits=4, out_dim=100, image_size=32):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim)
        self.n_exits = n_exits
        self.out_dim = out_dim


class ResNet18Base(ResNet):
    def __init__(self, n_exits=1, out_dim=100):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim)
        self.n_exits = n_exits
        self.out_dim = out_dim

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        out = self.bn1(self.conv1(x))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        return [self.linear(self._final(out))]


def _insert_stage_dropout(model):
    """Insertion rules of SA/models/resnet18/resnet18.py:224-240 / :272-288."""
    layer_list = [model.layer1, model.layer2, model.layer3, model.layer4]
    if model.dropout == "block":
        for i in range(len(layer_list) - 1):
            if model.mask_type == "mc":
                layer_list[i] = nn.Sequential(layer_list[i], MCDropout(model.dropout_p))
            else:
                layer_list[i] = nn.Sequential(
                    layer_list[i], Masksembles2D(layer_list[i][-1].planes, model.num_masks, model.mask_scale))
        model.layer1, model.layer2, model.layer3, model.layer4 = layer_list
    elif model.dropout == "layer":
        for b in range(len(layer_list)):
            for l in range(len(layer_list[b])):
                if b == len(layer_list) - 1 and l == len(layer_list[b]) - 1:
                    continue
                if model.mask_type == "mc":
                    layer_list[b][l] = nn.Sequential(layer_list[b][l], MCDropout(model.dropout_p))
                else:
                    # the reference dereferences an undefined loop variable here (:240, :288)
                    raise UnboundLocalError("dropout='layer' with mask_type='mask' is broken in the reference")


class ResNet18MC(ResNet):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=1, out_dim=100, image_size=32,
                 mask_type="mc", num_masks=4, mask_scale=4.0):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim)
        self.n_exits = n_exits
        self.out_dim = out_dim
        self.dropout_exit = dropout_exit
        self.dropout = dropout
        self.dropout_p = dropout_p
        self.mask_type = mask_type
        self.num_masks = num_masks
        self.mask_scale = mask_scale
        _insert_stage_dropout(self)
        if self.dropout_exit:
            if self.mask_type == "mc":
                self.exit_dropout = MCDropout(self.dropout_p)
            else:
                self.exit_dropout = Masksembles1D(512 * BasicBlock.expansion, self.num_masks, self.mask_scale)
        self._attach_ctx()

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        out = self.bn1(self.conv1(x))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        out = self._final(out)
        if self.dropout_exit:
            out = self.exit_dropout(out)
        return [self.linear(out)]


class ResNet18MCEarlyExit(ResNet):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=4, out_dim=100, image_size=32,
                 mask_type="mc", num_masks=4, mask_scale=4.0):
        super().__init__(block=BasicBlock, num_blocks=[2, 2, 2, 2], num_classes=out_dim)
        self.n_exits = n_exits
        self.out_dim = out_dim
        self.dropout_exit = dropout_exit
        self.dropout = dropout
        self.dropout_p = dropout_p
        self.mask_type = mask_type
        self.num_masks = num_masks
        self.mask_scale = mask_scale
        _insert_stage_dropout(self)
        if self.dropout_exit:
            if self.mask_type == "mc":
                self.exit1_dropout = MCDropout(self.dropout_p)
                self.exit2_dropout = MCDropout(self.dropout_p)
                self.exit3_dropout = MCDropout(self.dropout_p)
                self.exit_dropout = MCDropout(self.dropout_p)
            else:
                self.exit1_dropout = Masksembles1D(512, self.num_masks, self.mask_scale)
                self.exit2_dropout = Masksembles1D(512, self.num_masks, self.mask_scale)
                self.exit3_dropout = Masksembles1D(512, self.num_masks, self.mask_scale)
                self.exit_dropout = Masksembles1D(512 * BasicBlock.expansion, self.num_masks, self.mask_scale)
        self._attach_ctx()

    def forward(self, x, seed=None, t=None):
        """SA/models/resnet18/resnet18.py:302-346."""
        self.mc.begin_forward(seed, t)
        out = self.bn1(self.conv1(x))
        out = self.layer1(out)
        out1 = self._exit1(out)
        if self.dropout_exit:
            out1 = self.exit1_dropout(out1)
        out1 = self.ex1linear(out1)
        out = self.layer2(out)
        out2 = self._exit2(out)
        if self.dropout_exit:
            out2 = self.exit2_dropout(out2)
        out2 = self.ex2linear(out2)
        out = self.layer3(out)
        out3 = self._exit3(out)
        if self.dropout_exit:
            out3 = self.exit3_dropout(out3)
        out3 = self.ex3linear(out3)
        out = self.layer4(out)
        out = self._final(out)
        if self.dropout_exit:
            out = self.exit_dropout(out)
        out = self.linear(out)
        return [out1, out2, out3, out]


def dict_drop(dic, *keys):
    """SA/utils.py:7-12."""
    return {k: v for k, v in dic.items() if k not in keys}


def get_res_net_18(ensemble, network_hyperparams):
    """SA/models/resnet18/resnet18_loader.py:4-16."""
    if ensemble == "early_exit" or ensemble is None:
        return ResNet18EarlyExit(**dict_drop(network_hyperparams, "call", "load_model", "resnet_type", "dropout",
                                             "dropout_exit", "dropout_p", "mask_type", "num_masks", "mask_scale"))
    elif ensemble == "mc":
        return ResNet18MC(**dict_drop(network_hyperparams, "call", "load_model", "resnet_type"))
    elif ensemble == "mc_early_exit":
        return ResNet18MCEarlyExit(**dict_drop(network_hyperparams, "call", "load_model", "resnet_type"))
