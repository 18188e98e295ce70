"""``nn2bnn`` for torch, behind the reference's interface (Hardware_Artifact/converter/pytorch/nn2bnn.py):

* ``_convert_model(model, p)`` (:32-45) — recursively wraps EVERY ``Linear / MaxPool{1,2,3}d / Conv1d`` in
  ``BayesianDropout``, every ``Conv2d`` in ``BayesianDropout2D`` and every ``Conv3d`` in ``BayesianDropout3D``
  (in place, returns the model; a bare layer is returned wrapped).
* ``MCDropout(model, nSamples=10, p=0.5)`` (:7-30) — converts on construction; training mode = one stochastic
  pass; eval mode = the mean of ``nSamples`` stochastic passes of the model's output (logits).
  The reference's eval branch cannot run as shipped (it prints ``pred`` before assigning it, :25-26, and the module
  does not import: ``test.ThreeLayerNet`` is missing, :5); this mirror implements the evident intent,
  ``sum(pred) / len(pred)``.

What the engine accepts after conversion:

* the package's own VGG-19 mirrors (``models.vgg19``): conv -> [site] -> BN -> ReLU, an elementwise site behind every
  MaxPool2d, the classifier's site on its logits (``build_vgg_graph``; pinned case: the reference's ``VGG19`` through its
  own ``_convert_model``);
* the package's own ResNet mirrors (``models.resnet18``: ``ResNet18Base``, the early-exit and MC variants) — the converter
  recurses through them exactly as the reference's recurses through its own (``_convert_model`` on ``ResNet18Base``,
  SA/models/resnet18/resnet18.py:189-204, is the pinned case): every conv of every BasicBlock, the 1x1 shortcut convs (which
  then keep their own launch instead of riding in conv2) and the exit-head convs get a per-(image, channel) site between conv
  and BatchNorm, every classifier an elementwise site on its logits (``engine.build_resnet_graph``);
* a (nested) ``nn.Sequential`` CNN on 3x32x32 inputs made of Conv2d [+BatchNorm2d] [+ReLU], MaxPool2d(2),
  AdaptiveAvgPool2d(1) / AvgPool2d over the whole map, Flatten, Linear (+ReLU), Dropout (identity in eval).

The mask of a wrapped Conv2d lands BEFORE its BatchNorm (an "inner" site of the C ABI, include/bayesnn_fpga_amd.h), the last
Linear's mask multiplies the logits.
* any other module whose hand-written ``forward`` ``torch.fx`` can trace and that is made of what the engine has — conv [+ BatchNorm]
  [+ residual add] [+ ReLU], MaxPool2d(2), a global average pool + flatten + Linear per output, hidden Linears — through
  ``fx_frontend.build_graph_fx`` (round 4; pinned case: the reference's ``_convert_model`` on a small hand-written residual net,
  tests/golden/converter_custom.npz).  Anything else raises TypeError naming the node.
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
        fam = getattr(self.model, "family", None)
        self.resnet = fam in ("resnet", "vgg")       # one of the package's own mirrors: its forward returns a LIST of logits
        self.multi_exit = bool(self.resnet and getattr(self.model, "multi_exit", True))
        self.n_exits = (4 if fam == "resnet" else 5) if self.multi_exit else 1
        self.traced = not self.resnet and not isinstance(self.model, nn.Sequential)
        if self.traced:
            # a hand-written forward: compiled through torch.fx (fx_frontend.py); it may return one logits tensor or a list of them
            from .fx_frontend import traced_outputs
            self.n_exits, self.resnet = traced_outputs(self.model)
            self.multi_exit = self.n_exits > 1
        linears = [m for m in self.model.modules() if isinstance(m, nn.Linear)]
        if not linears:
            raise TypeError("the engine needs a model that ends in nn.Linear")
        self.out_dim = linears[-1].out_features
        self._init_engine_state()

    def extra_repr(self):
        return "nSamples: {}\nprobability: {}".format(self.nSamples, self.p)

    def forward(self, x):
        """Training mode: one stochastic pass; eval mode: the mean of ``nSamples`` passes.  A converted ResNet mirror returns
        what its reference forward returns — a LIST of logits, one per exit (``[out]`` for ``ResNet18Base``, :204) —, a
        converted Sequential the logits tensor."""
        if self.training:
            out = EngineModelMixin.forward(self, x)
            return out if self.resnet else out[0]
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("bayesnn_fpga_amd models run on an MI355X through the HIP engine; got a CPU tensor "
                               "(there is no CPU fallback)")
        eng = self.engine(x.device, max_batch=x.shape[0], calib=x)
        r = eng.predict(x, self.nSamples, seed=self.mc_seed, t_begin=self.mc_pass)
        self.advance(self.nSamples)
        mean = [r["logit_mean"][e].to(torch.float32) for e in range(self.n_exits)]
        return mean if self.resnet else mean[0]

    # ---- graph of the converted model (engine.GraphBuilder) ------------------------------------------------
    def build_graph(self, g):
        if getattr(self.model, "family", None) == "resnet":
            from ...engine import build_resnet_graph
            return build_resnet_graph(self.model, g)
        if getattr(self.model, "family", None) == "vgg":
            from ...models.vgg19.vgg19 import build_vgg_graph
            return build_vgg_graph(self.model, g)
        if not isinstance(self.model, nn.Sequential):
            from .fx_frontend import build_graph_fx
            n, _ = build_graph_fx(self.model, g)
            assert n == self.n_exits
            return
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
