"""VGG-19-BN multi-exit family, restated (TEST ORACLE — see oracle/__init__.py).

Follows SA/models/vgg19/vgg19.py: ``make_layers`` :121-143, ``make_classifier`` :146-183 (32-px branch),
``VGG`` :88-119, ``VGG19`` :186-192, ``VGG19MC`` :194-252, ``VGG19EarlyExit`` :256-324,
``VGG19MCEarlyExit`` :327-382.  Only the configurations that work in the reference are restated:
``dropout=None`` with optional exit dropout (every block/layer insertion variant raises at
construction in the reference); image_size 32.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import MCContext, MCDropout, Masksembles1D

CFG19 = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def make_layers(cfg):
    blocks, layers = nn.ModuleList(), nn.ModuleList()
    cin = 3
    for l in cfg:
        if l == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            blocks.append(layers)
            layers = nn.ModuleList()
            continue
        layers.append(nn.Conv2d(cin, l, kernel_size=3, padding=1))
        layers.append(nn.BatchNorm2d(l))
        layers.append(nn.ReLU(inplace=True))
        cin = l
    seq = nn.ModuleList()
    for b in blocks:
        seq.append(nn.Sequential(*b))
    return seq, blocks


def make_classifier(num_classes, mc_dropout_p=0, mask_type='mask', num_masks=4, mask_scale=4.0):
    if mc_dropout_p == 0:
        mods = [nn.Linear(512, num_classes)]
    elif mask_type == 'mc':
        mods = [MCDropout(p=mc_dropout_p), nn.Linear(512, num_classes)]
    else:
        mods = [Masksembles1D(512, num_masks, mask_scale), nn.Linear(512, num_classes)]
    return nn.Sequential(*mods)


class VGG(nn.Module):
    def __init__(self, blocks, num_class=100, image_size=32):
        super().__init__()
        self.blocks, self.non_sequentialized_blocks = blocks
        self.image_size = image_size
        self.avg_pool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = make_classifier(num_class, mask_type=None)
        self.init_weights()
        self.mc = MCContext()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.fill_(0.01)

    def _attach_ctx(self):
        for m in self.modules():
            if isinstance(m, MCDropout):
                m.ctx = self.mc

    def forward(self, x, seed=None, t=None):
        self.mc.begin_forward(seed, t)
        out = x
        for block in self.blocks:
            out = block(out)
        return [self.classifier(out.view(out.size(0), -1))]


class VGG19(VGG):
    def __init__(self, n_exits=1, out_dim=100, image_size=32):
        super().__init__(make_layers(CFG19), num_class=out_dim, image_size=image_size)
        self.n_exits, self.out_dim = n_exits, out_dim
        self.init_weights()


def _reject_block_dropout(dropout):
    if dropout is not None:
        # reference: AttributeError at construction (vgg19.py:224/235, :365/376)
        raise AttributeError("VGG block/layer dropout insertion is broken in the reference")


class VGG19MC(VGG19):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=1, out_dim=100, mask_type="mc",
                 num_masks=4, mask_scale=4.0, image_size=32):
        super().__init__(image_size=image_size)            # NB: reference builds the parent with its DEFAULT out_dim
        self.n_exits, self.out_dim = n_exits, out_dim
        self.dropout, self.dropout_p, self.dropout_exit = dropout, dropout_p, dropout_exit
        self.mask_type, self.num_masks, self.mask_scale = mask_type, num_masks, mask_scale
        if self.dropout_exit:
            self.classifier = make_classifier(self.out_dim, self.dropout_p, self.mask_type, self.num_masks, self.mask_scale)
        self.init_weights()
        _reject_block_dropout(dropout)
        self._attach_ctx()


class VGG19EarlyExit(VGG19):
    def __init__(self, n_exits=1, out_dim=100, image_size=32):
        super().__init__(n_exits=n_exits, out_dim=out_dim, image_size=image_size)
        for e, chans in ((1, (64, 128, 256, 512)), (2, (128, 256, 512)), (3, (256, 512))):
            convs = [nn.Conv2d(a, b, kernel_size=3, stride=2, padding=1, bias=False) for a, b in zip(chans[:-1], chans[1:])]
            bns = [nn.BatchNorm2d(b) for b in chans[1:]]
            relus = [nn.ReLU(inplace=True) for _ in chans[1:]]
            mods = []
            for c, b, r in zip(convs, bns, relus):
                mods += [c, b, r]
            setattr(self, f"ex{e}featureextractor", nn.Sequential(*mods))
            setattr(self, f"ex{e}linear", make_classifier(self.out_dim, mask_type=None))
        self.ex4linear = make_classifier(self.out_dim, mask_type=None)
        self.init_weights()

    def forward(self, x, seed=None, t=None):
        """vgg19.py:290-324."""
        self.mc.begin_forward(seed, t)
        outs = []
        out = self.blocks[0](x)
        for e, fe, lin in ((1, self.ex1featureextractor, self.ex1linear), (2, self.ex2featureextractor, self.ex2linear),
                           (3, self.ex3featureextractor, self.ex3linear)):
            o = F.avg_pool2d(fe(F.relu(out)), 2)
            outs.append(lin(o.view(o.size(0), -1)))
            out = self.blocks[e](out)
        o4 = F.avg_pool2d(out, 2)
        outs.append(self.ex4linear(o4.view(o4.size(0), -1)))
        out = self.blocks[4](out)
        outs.append(self.classifier(out.view(out.size(0), -1)))
        return outs


class VGG19MCEarlyExit(VGG19EarlyExit):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=4, out_dim=100, mask_type="mc",
                 num_masks=4, mask_scale=4.0, image_size=32):
        super().__init__(n_exits=n_exits, out_dim=out_dim, image_size=image_size)
        self.n_exits, self.out_dim = n_exits, out_dim
        self.dropout, self.dropout_p, self.dropout_exit = dropout, dropout_p, dropout_exit
        self.mask_type, self.num_masks, self.mask_scale = mask_type, num_masks, mask_scale
        if self.dropout_exit:
            mk = lambda: make_classifier(self.out_dim, self.dropout_p, self.mask_type, self.num_masks, self.mask_scale)
            self.ex1linear, self.ex2linear, self.ex3linear, self.ex4linear = mk(), mk(), mk(), mk()
            self.classifier = mk()
        self.init_weights()
        _reject_block_dropout(dropout)
        self._attach_ctx()
