"""The reference's torch "nn2bnn" converter, restated (TEST ORACLE — see oracle/__init__.py).

``convert_model`` follows ``_convert_model`` (Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45): every
Linear / MaxPool{1,2,3}d / Conv1d is followed by an always-on elementwise dropout (``BayesianDropout``,
Dropouts.py:25-34: ``F.dropout(layer(x), p, True)``), every Conv2d by a per-(image, channel) dropout
(``BayesianDropout2D`` :36-45, ``F.dropout2d``), Conv3d likewise (:47-56).  The Bernoulli source is the shared
Philox convention (oracle/philox.py) instead of ATen's; sites are numbered in call order.
``ConvertedNet`` is the wrapper of nn2bnn.py:7-30 in the calling convention of oracle/mcd.py: one stochastic pass
per call, the model's logits as a one-element list (the wrapper's eval-mode mean over nSamples is
``mcd.mcd_predict(...)["logit_mean"][0]``).
Pinned by tests/golden/converter_cnn.npz and converter_resnet18base.npz (the reference's own ResNet18Base, SA/models/resnet18/
resnet18.py:189-204, through its own converter), produced by the reference's own Dropouts.py + nn2bnn._convert_model.
"""
from torch import nn

from .layers import MCContext, philox_dropout

_ELEMENTWISE = (nn.Linear, nn.MaxPool1d, nn.MaxPool2d, nn.MaxPool3d, nn.Conv1d)
_CHANNELWISE = (nn.Conv2d, nn.Conv3d)


class _Wrapped(nn.Module):
    def __init__(self, layer, p, channelwise, ctx):
        super().__init__()
        if p < 0 or p > 1:
            raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(p))
        self.layer, self.p, self.channelwise, self.ctx = layer, p, channelwise, ctx

    def forward(self, x):
        return philox_dropout(self.ctx, self.layer(x), self.p, channelwise=self.channelwise)


def convert_model(model, p, ctx):
    if type(model) in _ELEMENTWISE:
        return _Wrapped(model, p, False, ctx)
    if type(model) in _CHANNELWISE:
        return _Wrapped(model, p, True, ctx)
    for name, child in model.named_children():
        setattr(model, name, convert_model(child, p, ctx))
    return model


class ConvertedNet(nn.Module):
    def __init__(self, model, p=0.5):
        super().__init__()
        self.ctx = MCContext()
        self.model = convert_model(model, p, self.ctx)
        self.p = p

    def forward(self, x, seed=None, t=None):
        self.ctx.begin_forward(seed, t)
        out = self.model(x)              # the restated ResNets return the list of logits their reference forward returns
        return out if isinstance(out, list) else [out]
