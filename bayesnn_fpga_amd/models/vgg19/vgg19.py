"""VGG-19-BN multi-exit family behind the reference's API (SA/models/vgg19/vgg19.py).

Same class names, constructor arguments, attributes and ``state_dict`` keys as the reference
(``VGG`` :88-119, ``VGG19`` :186-192, ``VGG19MC`` :194-252, ``VGG19EarlyExit`` :256-324,
``VGG19MCEarlyExit`` :327-382; factory ``get_vgg_19`` :16-42), same construction order (so a torch
seed gives the reference's initial weights).  As in the reference only exit dropout works: any
``dropout="block"/"layer"`` raises ``AttributeError`` at construction (reference :224/:235, :365/:376),
and the 224-px ImageNet variant is not on the accelerated path.  Parameter containers + graph
description; the arithmetic runs in the HIP engine (``build_vgg_graph``).
"""
import math

from torch import nn

from ...utils import Masksembles1D, dict_drop
from .._engine_mixin import EngineModelMixin
from ..resnet18.resnet18 import MCDropout

CFG19 = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
_EXIT_CHANNELS = {1: (64, 128, 256, 512), 2: (128, 256, 512), 3: (256, 512)}


def make_layers(cfg, batch_norm=True):
    raw, cur, cin = nn.ModuleList(), nn.ModuleList(), 3
    for item in cfg:
        if item == 'M':
            cur.append(nn.MaxPool2d(kernel_size=2, stride=2))
            raw.append(cur)
            cur = nn.ModuleList()
            continue
        cur.append(nn.Conv2d(cin, item, kernel_size=3, padding=1))
        if batch_norm:
            cur.append(nn.BatchNorm2d(item))
        cur.append(nn.ReLU(inplace=True))
        cin = item
    return nn.ModuleList(nn.Sequential(*b) for b in raw), raw


def make_classifier(size, num_classes, mc_dropout_p=0, mask_type='mask', num_masks=4, mask_scale=4.0):
    if size == 224:
        raise NotImplementedError("the 224-px VGG classifier is not on the accelerated path")
    if mc_dropout_p == 0:
        return nn.Sequential(nn.Linear(512, num_classes))
    site = MCDropout(p=mc_dropout_p) if mask_type == 'mc' else Masksembles1D(512, num_masks, mask_scale)
    return nn.Sequential(site, nn.Linear(512, num_classes))


class VGG(EngineModelMixin, nn.Module):
    family = "vgg"
    multi_exit = False

    def __init__(self, blocks, num_class=100, image_size=32):
        super().__init__()
        self.blocks, self.non_sequentialized_blocks = blocks
        self.image_size = image_size
        self.avg_pool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = make_classifier(image_size, num_class, mask_type=None)
        self.init_weights()
        self._init_engine_state()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.fill_(0.01)


class VGG19(VGG):
    def __init__(self, n_exits=1, out_dim=100, *args, **kwargs):
        super().__init__(make_layers(CFG19, batch_norm=True), num_class=out_dim, *args, **kwargs)
        self.n_exits, self.out_dim = n_exits, out_dim
        self.init_weights()


def _configure(model, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale):
    model.n_exits, model.out_dim = n_exits, out_dim
    model.dropout, model.dropout_p, model.dropout_exit = dropout, dropout_p, dropout_exit
    model.mask_type, model.num_masks, model.mask_scale = mask_type, num_masks, mask_scale


def _classifier(model):
    return make_classifier(model.image_size, model.out_dim, model.dropout_p, model.mask_type, model.num_masks, model.mask_scale)


def _reject_block_dropout(dropout):
    if dropout is not None:
        raise AttributeError("VGG block/layer dropout insertion is broken in the reference (vgg19.py:224,235,365,376)")


class VGG19MC(VGG19):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=1, out_dim=100, mask_type="mc",
                 num_masks=4, mask_scale=4.0, *args, **kwargs):
        super().__init__(*args, **kwargs)          # reference quirk: the parent is built with ITS default out_dim
        _configure(self, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale)
        if dropout_exit:
            self.classifier = _classifier(self)
        self.init_weights()
        _reject_block_dropout(dropout)


class VGG19EarlyExit(VGG19):
    multi_exit = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for e, ch in _EXIT_CHANNELS.items():
            convs = [nn.Conv2d(a, b, kernel_size=3, stride=2, padding=1, bias=False) for a, b in zip(ch[:-1], ch[1:])]
            bns = [nn.BatchNorm2d(b) for b in ch[1:]]
            relus = [nn.ReLU(inplace=True) for _ in ch[1:]]
            setattr(self, f"ex{e}featureextractor", nn.Sequential(*[m for trio in zip(convs, bns, relus) for m in trio]))
            setattr(self, f"ex{e}linear", make_classifier(self.image_size, self.out_dim, mask_type=None))
        self.ex4linear = make_classifier(self.image_size, self.out_dim, mask_type=None)
        self.init_weights()


class VGG19MCEarlyExit(VGG19EarlyExit):
    def __init__(self, dropout_exit=False, dropout=None, dropout_p=0.5, n_exits=4, out_dim=100, mask_type="mc",
                 num_masks=4, mask_scale=4.0, *args, **kwargs):
        super().__init__(n_exits=n_exits, out_dim=out_dim, *args, **kwargs)
        _configure(self, dropout_exit, dropout, dropout_p, n_exits, out_dim, mask_type, num_masks, mask_scale)
        if dropout_exit:
            self.ex1linear, self.ex2linear, self.ex3linear, self.ex4linear = (_classifier(self) for _ in range(4))
            self.classifier = _classifier(self)
        self.init_weights()
        _reject_block_dropout(dropout)


def get_vgg_19(ensemble, network_hyperparams):
    """SA/models/vgg19/vgg19.py:16-42."""
    base = dict_drop(network_hyperparams, "call", "load_model", "resnet_type")
    plain = dict_drop(base, "dropout", "dropout_exit", "dropout_p", "mask_type", "num_masks", "mask_scale")
    if ensemble is None:
        return VGG19(**plain)
    if ensemble == "early_exit":
        return VGG19EarlyExit(**plain)
    if ensemble == "mc":
        return VGG19MC(**base)
    if ensemble == "mc_early_exit":
        return VGG19MCEarlyExit(**base)
    raise ValueError


def _split_classifier(seq):
    """make_classifier output -> (site module | None, Linear)."""
    mods = list(seq)
    return (mods[0], mods[1]) if len(mods) == 2 else (None, mods[0])


def build_vgg_graph(model, g):
    """Op sequence of VGG19EarlyExit.forward (reference :290-324) / VGG.forward (:107-119).

    Also compiles a mirror that went through the torch converter (``nn2bnn._convert_model``, Hardware_Artifact/converter/
    pytorch/nn2bnn.py:32-45): a wrapped Conv2d carries a per-(image, channel) site between the conv and its BatchNorm, a wrapped
    MaxPool2d an elementwise site on its output, a wrapped classifier an elementwise site on its logits; site ids in call order."""
    from ...engine import _converted
    x = g.tensor(32, 32, 3)
    multi = model.multi_exit
    n_out = 5 if multi else 1
    first = True

    def conv(x, wrapped, bn, stem=False):
        m, w = _converted(wrapped)
        return g.conv(x, m, bn, relu=True, stem=stem, site=g.site(w, channelwise=True), site_inner=w is not None)

    def head(y, classifier, index):
        site, lin = _split_classifier(classifier)
        lin, w = _converted(lin)
        if w is not None and site is not None:
            raise TypeError("a converted classifier (dropout on the logits) on top of an exit dropout is not on the accelerated path")
        if w is not None:
            g.head(y, lin, index, site=g.site(w), site_on_logits=True)
        else:
            g.head(y, lin, index, site=g.site(site))

    for bi, block in enumerate(model.blocks):
        mods = list(block)
        i = 0
        while i < len(mods):
            m, w = _converted(mods[i])
            if isinstance(m, nn.Conv2d):
                bn = mods[i + 1] if isinstance(mods[i + 1], nn.BatchNorm2d) else None
                if w is not None and bn is None:
                    raise TypeError("a converted conv needs its BatchNorm (the site sits between the two)")
                x = conv(x, mods[i], bn, stem=first)             # conv(+bias) -> [site] -> BN -> ReLU
                first = False
                i += 3 if bn is not None else 2
            elif isinstance(m, nn.MaxPool2d):
                x = g.maxpool(x)
                if w is not None:
                    x = g.mask(x, g.site(w))
                i += 1
            else:
                raise TypeError(f"unexpected module {type(m).__name__} in a VGG block")
        if multi and bi < 3:
            # exit head bi+1: relu (idempotent) -> [conv s2 + BN + ReLU]* -> avg_pool2d(.,2) on the 2x2 map -> classifier
            y = x
            fe = list(getattr(model, f"ex{bi + 1}featureextractor"))
            for j in range(0, len(fe), 3):
                y = conv(y, fe[j], fe[j + 1])
            head(y, getattr(model, f"ex{bi + 1}linear"), bi)
        elif multi and bi == 3:
            head(x, model.ex4linear, 3)                           # avg_pool2d(out, 2) on the 2x2 block-4 output
    head(x, model.classifier, n_out - 1)
