"""``torch.fx`` front end of the converter: compiles a (converted) CNN with a HAND-WRITTEN ``forward`` into the engine's graph.

The reference's ``_convert_model`` recurses through any ``nn.Module`` (Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45): the
model it returns keeps whatever ``forward`` its author wrote.  The package's own ResNet / VGG mirrors and nested ``nn.Sequential``s
are compiled by their own builders (``engine.build_resnet_graph``, ``build_vgg_graph``, ``nn2bnn.MCDropout.build_graph``); anything
else is symbolically traced here — the dropout wrappers and the ``torch.nn`` layers as leaves — and the traced graph is matched,
in call order, against what the HIP engine has:

    conv      Conv2d [-> BatchNorm2d] [-> + other tensor] [-> relu]         (BMI_OP_STEM on the network input, BMI_OP_CONV after)
    maxpool   MaxPool2d(2, 2) / F.max_pool2d(x, 2)
    dense     Linear [-> relu] on a flattened 1x1 map                        (BMI_OP_DENSE)
    head      relu -> global average pool -> flatten -> Linear               (BMI_OP_HEAD; the model's outputs: one, or a list)
    site      a BayesianDropout* wrapper around a Conv2d (per (image, channel), between the conv and its BatchNorm), a MaxPool2d
              or a Linear (elementwise; on the logits for a classifier); nn.Dropout modules are identities in eval mode

Site ids follow the order in which the wrappers are CALLED (the order of the traced nodes), as the reference's forward would draw
them.  Whatever does not match raises ``TypeError`` naming the node — the engine has no generic fallback by design (there is no
CPU path in this package).
"""
import operator

import torch
import torch.fx
import torch.nn.functional as F
from torch import nn

from .Dropouts import BayesianDropout, BayesianDropout2D, BayesianDropout3D

_WRAP = (BayesianDropout, BayesianDropout2D, BayesianDropout3D)
_RELU_FUNCS = {F.relu, torch.relu, torch.relu_}
_FLATTEN_FUNCS = {torch.flatten}


class _Tracer(torch.fx.Tracer):
    def is_leaf_module(self, m, qualname):
        return isinstance(m, _WRAP) or m.__class__.__module__.startswith("torch.nn") and not isinstance(m, nn.Sequential)


def _unwrap(m):
    return (m.layer, m) if isinstance(m, _WRAP) else (m, None)


def traced_outputs(model):
    """(number of outputs, whether the forward returns a list / tuple) of a traceable forward; TypeError otherwise."""
    try:
        graph = _Tracer().trace(model)
    except Exception as exc:
        raise TypeError(f"{type(model).__name__}: the forward cannot be traced by torch.fx ({exc})") from exc
    out = [n for n in graph.nodes if n.op == "output"][0].args[0]
    return (len(out), True) if isinstance(out, (list, tuple)) else (1, False)


def build_graph_fx(model, g):
    """Emits ``model``'s forward into GraphBuilder ``g``.  Returns (n_exits, returns_list)."""
    try:
        graph = _Tracer().trace(model)
    except Exception as exc:               # data-dependent control flow, unsupported python, ...
        raise TypeError(f"{type(model).__name__}: the forward cannot be traced by torch.fx ({exc}); the engine compiles the package's own "
                        "ResNet / VGG mirrors, nested nn.Sequential CNNs and traceable forwards made of conv / BatchNorm / ReLU / add / "
                        "max-pool / global average pool / flatten / Linear") from exc
    mods = dict(model.named_modules())
    nodes = list(graph.nodes)
    users = {n: list(n.users) for n in nodes}
    val = {}            # fx node -> ("t", tensor id, relu_last) | ("flat", tensor id, relu_last) | ("pooled", tensor id) | ("exit", index) | ("size",)
    consumed = set()    # nodes folded into a previous op
    n_exits = [0]

    def bad(n, why):
        raise TypeError(f"{type(model).__name__}: node '{n.name}' ({n.op} {n.target}) is not on the accelerated path: {why}")

    def module_of(n):
        return mods[n.target] if n.op == "call_module" else None

    def is_relu(n):
        return (n.op == "call_module" and isinstance(module_of(n), nn.ReLU)) or (n.op == "call_function" and n.target in _RELU_FUNCS) or \
               (n.op == "call_method" and n.target in ("relu", "relu_"))

    def sole_user(n):
        return users[n][0] if len(users[n]) == 1 else None

    def tensor_arg(n, i=0):
        a = n.args[i]
        if not isinstance(a, torch.fx.Node) or a not in val:
            bad(n, "its input is not a tensor the engine produced")
        return val[a]

    def map_arg(n, what):
        """(kind, tensor id, relu_last) of a node that takes a feature map or a flattened one; a classifier's logits, a pooled vector or
        a .size() going in is a TypeError naming the node (not an unpacking error)."""
        v = tensor_arg(n)
        if v[0] not in ("t", "flat"):
            bad(n, f"{what} takes a feature map, not {'the logits of a classifier' if v[0] == 'exit' else 'a pooled vector' if v[0] == 'pooled' else 'a size'}")
        return v

    first = [True]
    for n in nodes:
        if n in consumed:
            continue
        if n.op == "placeholder":
            if any(v[0] == "t" for v in val.values()):
                bad(n, "one network input only")
            val[n] = ("t", g.tensor(32, 32, 3), False)
            continue
        if n.op == "output":
            outs = n.args[0]
            returns_list = isinstance(outs, (list, tuple))
            outs = list(outs) if returns_list else [outs]
            idx = []
            for o in outs:
                if not isinstance(o, torch.fx.Node) or val.get(o, ("",))[0] != "exit":
                    bad(n, "every output must be the logits of a classifier (… -> relu -> global average pool -> flatten -> Linear)")
                idx.append(val[o][1])
            if idx != list(range(len(idx))):
                bad(n, "the outputs must be returned in the order the classifiers are called")
            return len(idx), returns_list
        m = module_of(n)
        layer, wrap = _unwrap(m) if m is not None else (None, None)
        # ---- conv [-> bn] [-> add] [-> relu] ----
        if isinstance(layer, nn.Conv2d):
            kind, x, _ = map_arg(n, "a convolution")
            if kind != "t":
                bad(n, "a convolution needs a feature map")
            cur, bn, res, relu = n, None, -1, False
            u = sole_user(cur)
            if u is not None and isinstance(module_of(u), nn.BatchNorm2d):
                bn, cur = module_of(u), u
                consumed.add(u)
                u = sole_user(cur)
            if u is not None and u.op == "call_function" and u.target in (operator.add, torch.add, operator.iadd) and len(u.args) == 2:
                if u.kwargs:
                    bad(u, "a residual add with keyword arguments (alpha != 1, out=) is not a plain add")
                other = u.args[1] if u.args[0] is cur else u.args[0]
                if isinstance(other, torch.fx.Node) and other in val and val[other][0] == "t":
                    res, cur = val[other][1], u
                    consumed.add(u)
                    u = sole_user(cur)
            if u is not None and is_relu(u):
                relu, cur = True, u
                consumed.add(u)
            if wrap is not None and bn is None and res >= 0:
                bad(n, "a converted conv with a residual needs its BatchNorm (the site sits between the two)")
            site = g.site(wrap, channelwise=True) if wrap is not None else None
            # without a BatchNorm the mask commutes with the ReLU (multipliers are >= 0): an ordinary outer site
            out = g.conv(x, layer, bn, relu=relu, residual=res, site=site, stem=first[0], site_inner=wrap is not None and bn is not None)
            if first[0] and res >= 0:
                bad(n, "the first convolution (3 input channels) takes no residual")
            first[0] = False
            val[cur] = ("t", out, relu)
            continue
        if isinstance(layer, nn.BatchNorm2d):
            bad(n, "a BatchNorm2d must directly follow the Conv2d it normalises (and be that conv's only reader)")
        # ---- max-pool ----
        is_fmax = n.op == "call_function" and n.target is F.max_pool2d
        if isinstance(layer, nn.MaxPool2d) or is_fmax:
            ks = layer.kernel_size if layer is not None else (n.args[1] if len(n.args) > 1 else n.kwargs.get("kernel_size"))
            st = layer.stride if layer is not None else (n.args[2] if len(n.args) > 2 else n.kwargs.get("stride", None))
            pd = layer.padding if layer is not None else n.kwargs.get("padding", 0)
            if ks not in (2, (2, 2)) or st not in (None, 2, (2, 2)) or pd not in (0, (0, 0)):
                bad(n, "only MaxPool2d(2, 2)")
            kind, x, rl = map_arg(n, "a max-pool")
            if kind != "t":
                bad(n, "a max-pool needs a feature map")
            x = g.maxpool(x)
            if wrap is not None:
                x = g.mask(x, g.site(wrap))
            val[n] = ("t", x, rl)
            continue
        # ---- relu on its own (e.g. F.relu(out) in front of an exit head: idempotent on a post-ReLU map) ----
        if is_relu(n):
            kind, x, rl = map_arg(n, "a ReLU")
            if not rl:
                bad(n, "a ReLU must directly follow its convolution / Linear (or repeat one)")
            val[n] = (kind, x, True)
            continue
        # ---- global average pool -> flatten -> Linear ----
        is_favg = n.op == "call_function" and n.target in (F.avg_pool2d, F.adaptive_avg_pool2d)
        if isinstance(layer, (nn.AdaptiveAvgPool2d, nn.AvgPool2d)) or is_favg:
            kind, x, rl = map_arg(n, "an average pool")
            h, w, _ = g.tensors[x]
            if isinstance(layer, nn.AdaptiveAvgPool2d) or (is_favg and n.target is F.adaptive_avg_pool2d):
                size = layer.output_size if layer is not None else (n.args[1] if len(n.args) > 1 else n.kwargs.get("output_size"))
                whole = size in (1, (1, 1))
            else:
                size = layer.kernel_size if layer is not None else (n.args[1] if len(n.args) > 1 else n.kwargs.get("kernel_size"))
                whole = size in (h, (h, w))
            if kind != "t" or not whole or not rl:
                bad(n, "only a global average pool behind a ReLU")
            val[n] = ("pooled", x)
            continue
        if n.op == "call_method" and n.target == "size":
            val[n] = ("size",)
            continue
        is_flat = (n.op == "call_method" and n.target in ("view", "reshape", "flatten")) or (n.op == "call_function" and n.target in _FLATTEN_FUNCS) or \
                  isinstance(layer, nn.Flatten)
        if is_flat:
            src = tensor_arg(n)
            if src[0] == "pooled":
                val[n] = src
            elif src[0] == "t" and g.tensors[src[1]][:2] == (1, 1):
                val[n] = ("flat", src[1], src[2])
            elif src[0] == "flat":
                val[n] = src
            else:
                bad(n, "Flatten of a map larger than 1x1 (pool it first)")
            continue
        if isinstance(layer, nn.Linear):
            src = tensor_arg(n)
            u = sole_user(n)
            to_output = not users[n] or all(x.op == "output" for x in users[n])
            if src[0] == "pooled" and not to_output:
                # (the engine pools inside its fused classifier kernel only: avgpool -> flatten -> fc1 -> relu -> fc2 has no op for fc1)
                bad(n, "a hidden Linear behind a global average pool: only a classifier (whose logits are returned) may read a pooled map")
            if src[0] in ("exit", "size"):
                bad(n, "a Linear takes a pooled or flattened feature vector")
            if src[0] == "pooled" or (src[0] == "flat" and to_output):
                if src[0] == "flat" and not src[2]:
                    bad(n, "the classifier input must come out of a ReLU")
                g.head(src[1], layer, n_exits[0], site=g.site(wrap), site_on_logits=True)
                val[n] = ("exit", n_exits[0])
                n_exits[0] += 1
                continue
            if src[0] != "flat":
                bad(n, "a hidden Linear needs a flattened 1x1 map")
            relu = u is not None and is_relu(u)
            if relu:
                consumed.add(u)
            out = g.dense(src[1], layer, relu=relu, site=g.site(wrap))        # Linear -> dropout -> ReLU == Linear -> ReLU -> dropout
            val[u if relu else n] = ("flat", out, relu)
            continue
        if isinstance(layer, (nn.Dropout, nn.Identity)):
            val[n] = tensor_arg(n)                                             # nn.Dropout is off in eval mode
            continue
        bad(n, "no engine op for it")
    raise TypeError(f"{type(model).__name__}: the traced forward returns nothing")
