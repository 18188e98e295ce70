"""``nn2bnn`` for torch, behind the reference's interface (Hardware_Artifact/converter/pytorch/nn2bnn.py):

* ``_convert_model(model, p)`` (:32-45) — recursively wraps EVERY ``Linear / MaxPool{1,2,3}d / Conv1d`` in
  ``BayesianDropout``, every ``Conv2d`` in ``BayesianDropout2D`` and every ``Conv3d`` in ``BayesianDropout3D``
  (in place, returns the model; a bare layer is returned wrapped).
* ``MCDropout(model, nSamples=10, p=0.5)`` (:7-30) — converts on construction; training mode = one stochastic
  pass; eval mode = the mean of ``nSamples`` stochastic passes of the model's output (logits).
  The reference's eval branch cannot run as shipped (it prints ``pred`` before assigning it, :25-26, and the module
  does not import: ``test.ThreeLayerNet`` is missing, :5); this mirror implements the evident intent,
  ``sum(pred) / len(pred)``.

What the engine accepts after conversion: a (nested) ``nn.Sequential`` CNN on 3x32x32 inputs made of
Conv2d [+BatchNorm2d] [+ReLU], MaxPool2d(2), AdaptiveAvgPool2d(1) / AvgPool2d over the whole map, Flatten, Linear
(+ReLU), Dropout (identity in eval).  The mask of a wrapped Conv2d lands BEFORE its BatchNorm (an "inner" site of the
C ABI, include/bayesnn_fpga_amd.h), the last Linear's mask multiplies the logits.  Anything else raises TypeError.
"""
import torch
from torch import nn

from ...models._engine_mixin import EngineModelMixin
from .Dropouts import BayesianDropout, BayesianDropout2D, BayesianDropout3D

_WRAPPERS = {nn.Linear: BayesianDropout, nn.MaxPool1d: BayesianDropout, nn.MaxPool2d: BayesianDropout,
             nn.MaxPool3d: BayesianDropout, nn.Conv1d: BayesianDropout, nn.Conv2d: BayesianDropout2D,
             nn.Conv3d: BayesianDropout3D}


def _convert_model(model, p):
    wrapper = _WRAPPERS.get(type(model))
    if wrapper is not None:
        return wrapper(model, p)
    for name, child in model.named_children():
        setattr(model, name, _convert_model(child, p))
    return model


def _leaves(module):
    """The converted model as the flat list of layers a nested nn.Sequential applies in order."""
    if isinstance(module, nn.Sequential):
        return [leaf for child in module for leaf in _leaves(child)]
    return [module]


def _unwrap(m):
    return (m.layer, m) if isinstance(m, (BayesianDropout, BayesianDropout2D, BayesianDropout3D)) else (m, None)


class MCDropout(EngineModelMixin, nn.Module):
    family = "converted"
    multi_exit = False

    def __init__(self, model, nSamples=10, p=0.5):
        super().__init__()
        self.model = _convert_model(model, p)
        self.nSamples = nSamples
        self.p = p
        self.n_exits = 1
        linears = [m for m in self.model.modules() if isinstance(m, nn.Linear)]
        if not linears:
            raise TypeError("the engine needs a model that ends in nn.Linear")
        self.out_dim = linears[-1].out_features
        self._init_engine_state()

    def extra_repr(self):
        return "nSamples: {}\nprobability: {}".format(self.nSamples, self.p)

    def forward(self, x):
        if self.training:
            return EngineModelMixin.forward(self, x)[0]
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("bayesnn_fpga_amd models run on an MI355X through the HIP engine; got a CPU tensor "
                               "(there is no CPU fallback)")
        eng = self.engine(x.device, max_batch=x.shape[0])
        r = eng.predict(x, self.nSamples, seed=self.mc_seed, t_begin=self.mc_pass)
        self.advance(self.nSamples)
        return r["logit_mean"][0].to(torch.float32)

    # ---- graph of the converted model (engine.GraphBuilder) ------------------------------------------------
    def build_graph(self, g):
        if not isinstance(self.model, nn.Sequential):
            raise TypeError("the engine compiles converted nn.Sequential CNNs; got " + type(self.model).__name__)
        mods = _leaves(self.model)
        x = g.tensor(32, 32, 3)
        first, flat, relu_last, i = True, False, False, 0
        while i < len(mods):
            m, wrap = _unwrap(mods[i])
            nxt = [type(_unwrap(k)[0]) for k in mods[i + 1:i + 3]] + [None, None]
            if isinstance(m, nn.Conv2d) and not flat:
                bn = mods[i + 1] if nxt[0] is nn.BatchNorm2d else None
                j = i + 1 + (bn is not None)
                relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                site = g.site(wrap, channelwise=True) if wrap is not None else None
                # without a BatchNorm the mask commutes with the ReLU (multipliers are >= 0): an ordinary outer site
                x = g.conv(x, m, bn, relu=relu, site=site, stem=first, site_inner=wrap is not None and bn is not None)
                first, relu_last, i = False, relu, j + relu
            elif isinstance(m, nn.MaxPool2d) and not flat:
                if m.kernel_size not in (2, (2, 2)) or m.stride not in (2, (2, 2)) or m.padding not in (0, (0, 0)):
                    raise TypeError("only MaxPool2d(2, 2) is on the accelerated path")
                x = g.maxpool(x)
                if wrap is not None:
                    x = g.mask(x, g.site(wrap))
                i += 1
            elif isinstance(m, (nn.AdaptiveAvgPool2d, nn.AvgPool2d)) and not flat:
                h, w, _ = g.tensors[x]
                whole = (m.output_size in (1, (1, 1))) if isinstance(m, nn.AdaptiveAvgPool2d) else (m.kernel_size in (h, (h, w)))
                if not whole or not relu_last:
                    raise TypeError("only a global average pool behind a ReLU is on the accelerated path")
                pooled, i = True, i + 1
                if not (i < len(mods) and isinstance(mods[i], nn.Flatten)):
                    raise TypeError("a global average pool must be followed by Flatten")
                m2, wrap2 = _unwrap(mods[i + 1]) if i + 1 < len(mods) else (None, None)
                if not isinstance(m2, nn.Linear) or i + 2 != len(mods):
                    raise TypeError("a global average pool must feed the final Linear")
                g.head(x, m2, 0, site=g.site(wrap2), site_on_logits=True)
                return
            elif isinstance(m, nn.Flatten):
                if g.tensors[x][:2] != (1, 1):
                    raise TypeError("Flatten of a map larger than 1x1 is not on the accelerated path (pool it first)")
                flat, i = True, i + 1
            elif isinstance(m, nn.Linear) and flat:
                if i + 1 == len(mods):                                   # classifier: dropout on the logits
                    if not relu_last:
                        raise TypeError("the classifier input must come out of a ReLU")
                    g.head(x, m, 0, site=g.site(wrap), site_on_logits=True)
                    return
                relu = nxt[0] is nn.ReLU
                # Linear -> dropout -> ReLU == Linear -> ReLU -> dropout
                x = g.dense(x, m, relu=relu, site=g.site(wrap))
                relu_last, i = relu, i + 1 + relu
            elif isinstance(m, (nn.Dropout, nn.Identity)):
                i += 1                                                   # nn.Dropout is off in eval mode
            else:
                raise TypeError(f"{type(m).__name__} at position {i} is not on the accelerated path")
        raise TypeError("the model must end in nn.Linear")
