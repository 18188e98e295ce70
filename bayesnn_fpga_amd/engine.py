"""Host side of the MI355X MCD engine: compiles a model of ``bayesnn_fpga_amd.models`` into the
C-ABI graph (include/bayesnn_fpga_amd.h), owns the device buffers, and drives the library.

PyTorch is plumbing here (device memory, streams, ``torch.distributed``); all arithmetic of the
path runs in ``libbayesnn_fpga_amd.so``.  What this replaces in the reference:

* ``GraphBuilder``       — the op sequence of ``ResNet18MCEarlyExit.forward``
                           (SA/models/resnet18/resnet18.py:302-346; single-exit :246-258, :195-204),
                           with eval-mode BatchNorm folded to per-channel scale/bias and the
                           stochastic layers turned into *sites* numbered in call order
                           (SURVEY.md Appendix C).
* ``MCDEngine.predict``  — ``FullAnalysis._get_output`` (SA/train/results_analyzer.py:236-270): T passes,
                           per-exit softmax, float64 mean over T (+ build-defined variance).
"""
import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from .utils import Masksembles1D, Masksembles2D

DEFAULT_CHUNK_IMAGES = 25600      # image-samples folded into one suffix launch (tuned on MI355X)
DEFAULT_CHUNK_SAMPLES_MAX = 128   # ... but never more than this many samples (the workspace is sized for a full chunk)


def _is_site(m):
    return isinstance(m, (nn.Dropout, Masksembles1D, Masksembles2D))


def fold_bn(bn, conv_bias=None):
    """eval-mode BatchNorm2d -> (scale, bias) fp32:  y = conv * scale + bias."""
    with torch.no_grad():
        scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        bias = bn.bias.double() - bn.running_mean.double() * scale
        if conv_bias is not None:
            bias = bias + conv_bias.double() * scale
    return scale.float(), bias.float()


class GraphBuilder:
    """Collects tensors / ops / device-resident weights for one model on one device."""

    def __init__(self, device, dtype="f16"):
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be 'f16', 'bf16', 'f32' (the exact engine), 'f16x2' or 'bf16x3' (the split engines), got {dtype!r}")
        self.device = device
        self.dtype = dtype
        # conv weights: the engine's 16-bit type, fp32 (exact engine), or 16-bit head + tail planes (split engines: conv_weight below)
        self.act_dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32, "f16x2": torch.float16, "bf16x3": torch.bfloat16}[dtype]
        self.tensors = []          # (h, w, c)
        self.ops = []              # dicts
        self.keep = []             # device tensors that must outlive the engine
        self.site_count = 0

    def tensor(self, h, w, c):
        self.tensors.append((int(h), int(w), int(c)))
        return len(self.tensors) - 1

    def dev(self, t, dtype):
        d = t.detach().to(device=self.device, dtype=dtype).contiguous()
        self.keep.append(d)
        return d

    def conv_weight(self, w, lift=None):
        """Device copy of a conv weight [Cout][ky][kx][Cin] in the engine's layout.  Split engines (BMI_DTYPE_F16X2 / BF16X3): the 16-bit
        head and tail planes [2][Cout][ky][kx][Cin], hi = rn16(w), lo = rn16(w - hi) — split once here, csrc/conv_split.hip reads both.
        ``lift`` [Cout] (``channel_lift``): the weights of output channel c are multiplied by the power of two lift[c] BEFORE they are
        rounded / split — exact — and the caller folds 1 / lift[c] into the channel's epilogue scale."""
        w32 = w.detach().float()
        if lift is not None:
            w32 = w32 * lift.to(w32.device).reshape((-1,) + (1,) * (w32.dim() - 1))
        if self.dtype not in ("f16x2", "bf16x3"):
            return self.dev(w32, self.act_dtype)
        hi = w32.to(self.act_dtype)
        lo = (w32 - hi.float()).to(self.act_dtype)
        return self.dev(torch.stack([hi, lo]), self.act_dtype)

    def channel_lift(self, *weights):
        """Per-output-channel power of two 2^k that brings max|w| of the channel (over ALL the given weights: a conv and its fused shortcut
        accumulate into one register) to [2^7, 2^8); None in the exact engine (fp32 weights).  Why: fp16 is denormal below 6.1e-5 and the
        split engines' fp16 TAIL rn16(w - hi) of any weight below 2^-3 is a subnormal (ulp 2^-24) — BN-folded conv weights of 1e-2 .. 1e-3
        would keep 15-18 of the 22 bits the split form is for, anything under 6e-8 nothing (round-5 advisor finding); the plain fp16 engine
        loses bits of every weight under 6.1e-5 the same way.  Lifted, every head and every tail that matters is a normal number; 2^-k goes
        into the fp32 epilogue scale — exact, and for weights that were normal numbers anyway bit for bit the unlifted result (a power of
        two commutes with every rounding on the way).  bf16 has fp32's exponent range: harmless there, one code path."""
        if self.dtype == "f32":
            return None
        amax = torch.stack([w.detach().float().abs().reshape(w.shape[0], -1).amax(dim=1) for w in weights]).amax(dim=0)
        k = 7 - torch.floor(torch.log2(amax.clamp_min(1e-30)))
        k = torch.where(amax > 0, k, torch.zeros_like(k)).clamp_(-8, 30)
        return torch.pow(2.0, k)

    def site(self, module, channelwise=False):
        """Allocates the next site id (call order) for a stochastic layer, or none."""
        if module is None:
            return None
        sid = self.site_count
        self.site_count += 1
        if channelwise:
            return dict(kind=_lib.SITE_CHANNEL, site_id=sid, p=float(module.p))
        if isinstance(module, (Masksembles1D, Masksembles2D)):
            masks = self.dev(module.masks, torch.float32)
            return dict(kind=_lib.SITE_MASKSEMBLE, site_id=sid, num_masks=module.n, masks=masks)
        return dict(kind=_lib.SITE_ELEMENTWISE, site_id=sid, p=float(module.p))

    def conv(self, x, conv, bn, relu, residual=-1, site=None, stem=False, shortcut=None, site_inner=False):
        """conv (+ folded BN) op.  ``shortcut=(x2, conv1x1, bn)`` fuses the BasicBlock downsample path into this
        conv as extra K-steps: both BN scales are folded into the fp16 weights, the biases are summed.
        ``site_inner``: the site sits between the conv and its BatchNorm (converter/pytorch rule):
        out = relu((conv*scale + scale*conv.bias) * mask + bn_shift)."""
        h, w, cin = self.tensors[x]
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out = self.tensor(ho, wo, conv.out_channels)
        if bn is not None:
            scale, bias = fold_bn(bn, conv.bias)
        else:
            scale = torch.ones(conv.out_channels)
            bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels)
        bias_post = None
        if site_inner and site is not None:
            if bn is None or shortcut is not None:
                raise ValueError("an inner site needs a BatchNorm behind the conv and no fused shortcut")
            scale, shift = fold_bn(bn, None)
            bias_post = shift
            bias = scale * conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(scale)
        wk = conv.weight.detach().permute(0, 2, 3, 1)          # [Cout][ky][kx][Cin]
        in2, w2dev = -1, None
        if shortcut is not None:
            x2, conv_s, bn_s = shortcut
            s_scale, s_bias = fold_bn(bn_s, conv_s.bias)
            wk = wk.float() * scale[:, None, None, None]
            w2 = conv_s.weight.detach().float()[:, :, 0, 0] * s_scale[:, None]       # [Cout][Cin2]
            # (the 16-bit engines' shortcut-carrying kernels take no scale beside in2 — both BN scales are in the weights — so no lift there)
            lift = self.channel_lift(wk, w2) if self.dtype in ("f16x2", "bf16x3") else None
            w2dev = self.conv_weight(w2, lift)                                         # (split engines: head / tail planes)
            bias = bias + s_bias
            scale = None if lift is None else (1.0 / lift).to(bias.device)
            in2 = x2
        else:
            lift = None if stem else self.channel_lift(wk)
            if lift is not None:
                scale = scale / lift.to(scale.device)
        wdev = self.dev(wk, torch.float32) if stem else self.conv_weight(wk, lift)
        self.ops.append(dict(kind=_lib.OP_STEM if stem else _lib.OP_CONV, in_=x, out=out, residual=residual, ksize=k,
                             stride=s, pad=p, relu=int(relu), weight=wdev, in2=in2, weight2=w2dev,
                             scale=self.dev(scale, torch.float32) if scale is not None else None,
                             bias=self.dev(bias, torch.float32), site=site,
                             bias_post=self.dev(bias_post, torch.float32) if bias_post is not None else None,
                             site_pos=_lib.SITE_POS_INNER if bias_post is not None else _lib.SITE_POS_OUTER))
        return out

    def mask(self, x, site):
        h, w, c = self.tensors[x]
        out = self.tensor(h, w, c)
        self.ops.append(dict(kind=_lib.OP_MASK, in_=x, out=out, residual=-1, site=site))
        return out

    def maxpool(self, x):
        h, w, c = self.tensors[x]
        out = self.tensor(h // 2, w // 2, c)
        self.ops.append(dict(kind=_lib.OP_MAXPOOL, in_=x, out=out, residual=-1))
        return out

    def dense(self, x, linear, relu, site=None):
        """Hidden fully-connected layer on a flattened [N,1,1,C] tensor, fp32 weights / accumulation / output
        (BMI_OP_DENSE: as a 1x1 fp16 conv the two VGG-11 dense layers pushed the predictive mean past 1e-3 at B=250)."""
        h, w, cin = self.tensors[x]
        if (h, w) != (1, 1) or cin != linear.in_features:
            raise ValueError("dense layers run on flattened [N,1,1,C] tensors")
        if cin % 32 or linear.out_features % 64:
            # (zero-padding would move the Philox element index of a site on the padded tensor — index = row * C + c — away from
            #  the reference's; the converter puts a site behind EVERY Linear)
            raise TypeError(f"hidden Linear({cin}, {linear.out_features}): the accelerated path takes in_features % 32 == 0 and "
                            "out_features % 64 == 0")
        out = self.tensor(1, 1, linear.out_features)
        self.ops.append(dict(kind=_lib.OP_DENSE, in_=x, out=out, residual=-1, relu=int(relu),
                             weight=self.dev(linear.weight, torch.float32), bias=self.dev(linear.bias, torch.float32), site=site))
        return out

    def head(self, x, linear, exit_index, site=None, site_on_logits=False):
        """relu -> global average pool -> [site] -> Linear -> [site on the logits] -> softmax."""
        c_in = self.tensors[x][2]
        if linear.in_features != c_in:
            raise ValueError(f"classifier expects {linear.in_features} features, pooled tensor has {c_in}")
        if c_in % 32:
            raise TypeError(f"classifier Linear({c_in}, {linear.out_features}): the accelerated path takes in_features % 32 == 0")
        cpad = (linear.out_features + 31) // 32 * 32
        w = torch.zeros(cpad, c_in)
        w[:linear.out_features] = linear.weight.detach().float()
        self.ops.append(dict(kind=_lib.OP_HEAD, in_=x, out=exit_index, residual=-1,
                             weight=self.dev(w, torch.float32), bias=self.dev(linear.bias, torch.float32), site=site,
                             site_pos=_lib.SITE_POS_INNER if (site_on_logits and site) else _lib.SITE_POS_OUTER))


def _unwrap(module):
    """Sequential(inner, site) wrappers made by the dropout-insertion rules -> (inner, site | None)."""
    if isinstance(module, nn.Sequential) and len(module) == 2 and _is_site(module[1]):
        return module[0], module[1]
    return module, None


def _can_fuse_shortcut(t_in, blk, dtype="f16"):
    """The 1x1 strided downsample conv rides along in the 3x3 patch kernel when that kernel takes conv2
    (16x16 / 8x8 / 4x4 maps, Cout % 128 == 0) and the block input has a multiple of 64 channels.
    BMI_FUSE_SHORTCUT=0 keeps the separate launch + residual (A/B, tests); the exact engine never fuses (the fusion folds
    the BN scales into the 16-bit weights: a speed feature)."""
    if os.environ.get("BMI_FUSE_SHORTCUT", "1") == "0" or dtype == "f32":
        return False
    h, w, c = t_in
    ds = blk.downsample[0]
    if dtype in _lib.FP32_ACT_DTYPES:       # the split engines: extra K-steps of the generic kernel, whatever the map (csrc/conv_split.hip)
        return c % 32 == 0 and ds.in_channels % 32 == 0 and ds.kernel_size == (1, 1) and ds.bias is None and ds.padding == (0, 0)
    return ((h, w) in ((16, 16), (8, 8), (4, 4)) and blk.conv2.out_channels % 128 == 0 and c % 64 == 0
            and ds.in_channels % 64 == 0 and ds.kernel_size == (1, 1) and ds.stride == (2, 2) and ds.bias is None)


def _converted(m):
    """``(layer, wrapper | None)`` for a layer the torch "nn2bnn" converter may have wrapped (converter/pytorch/Dropouts.py:
    BayesianDropout* own the layer as ``.layer`` and drop its OUTPUT in every mode)."""
    if m is not None and type(m).__name__.startswith("BayesianDropout") and hasattr(m, "layer"):
        return m.layer, m
    return m, None


def build_resnet_graph(model, g):
    """Op sequence of the ResNet family forwards (reference resnet18.py:302-346 / :246-258 / :195-204).

    Also compiles a mirror that went through the converter (``nn2bnn._convert_model``: every Conv2d wrapped in
    BayesianDropout2D, every Linear in BayesianDropout — Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45): a wrapped conv
    carries a per-(image, channel) site BETWEEN the conv and its BatchNorm (an inner site of the C ABI), a wrapped Linear an
    elementwise site on its logits; site ids follow the CALL order of the reference forward (BasicBlock.forward :32-48: conv1,
    conv2, then the downsample conv), and a shortcut conv that carries a site keeps its own launch."""
    x = g.tensor(32, 32, 3)                                   # tensor 0: network input (fp32 NCHW)
    stem, stem_w = _converted(model.conv1)
    x = g.conv(x, stem, model.bn1, relu=False, stem=True,       # no ReLU after the stem (:303)
               site=g.site(stem_w, channelwise=True), site_inner=stem_w is not None)
    multi = getattr(model, "multi_exit", True)
    dropout_exit = getattr(model, "dropout_exit", False)
    exit_sites = {1: "exit1_dropout", 2: "exit2_dropout", 3: "exit3_dropout"}

    def head(y, linear, index, feature_site_module):
        lin, lin_w = _converted(linear)
        if lin_w is not None and feature_site_module is not None:
            raise TypeError("a converted classifier (dropout on the logits) on top of an exit dropout is not on the accelerated path")
        if lin_w is not None:
            g.head(y, lin, index, site=g.site(lin_w), site_on_logits=True)
        else:
            g.head(y, lin, index, site=g.site(feature_site_module))

    for si in range(1, 5):
        stage, stage_site = _unwrap(getattr(model, f"layer{si}"))
        blocks = list(stage)
        for bi, blk in enumerate(blocks):
            blk, blk_site = _unwrap(blk)
            site_mod = blk_site if blk_site is not None else (stage_site if bi == len(blocks) - 1 else None)
            c1, w1 = _converted(blk.conv1)
            c2, w2 = _converted(blk.conv2)
            ds, wd = _converted(blk.downsample[0]) if blk.downsample is not None else (None, None)
            if w2 is not None and site_mod is not None:
                raise TypeError("a converted conv (site before its BatchNorm) under a block / stage dropout is not on the accelerated path")
            a = g.conv(x, c1, blk.bn1, relu=True, site=g.site(w1, channelwise=True), site_inner=w1 is not None)
            # site ids follow call order: a block's site is allocated when the block finishes; conv2's before the shortcut's
            site2 = g.site(w2, channelwise=True) if w2 is not None else g.site(site_mod)
            if ds is not None and wd is None and w2 is None and _can_fuse_shortcut(g.tensors[a], blk, g.dtype):
                x = g.conv(a, c2, blk.bn2, relu=True, site=site2, shortcut=(x, ds, blk.downsample[1]))
            else:
                res = x
                if ds is not None:
                    res = g.conv(x, ds, blk.downsample[1], relu=False, site=g.site(wd, channelwise=True), site_inner=wd is not None)
                x = g.conv(a, c2, blk.bn2, relu=True, residual=res, site=site2, site_inner=w2 is not None)
        if multi and si < 4:
            # exit head si: relu -> conv s2 -> bn chain, relu, avg-pool, [exit dropout], linear
            # (F.relu on a stage output is idempotent: it is already >= 0 and masks keep the sign)
            y = x
            n_conv = 4 - si
            for j in range(1, n_conv + 1):
                ec, ew = _converted(getattr(model, f"ex{si}conv{j}"))
                y = g.conv(y, ec, getattr(model, f"ex{si}bn{j}"), relu=True, site=g.site(ew, channelwise=True), site_inner=ew is not None)
            sm = getattr(model, exit_sites[si], None) if dropout_exit else None
            head(y, getattr(model, f"ex{si}linear"), si - 1, sm)
    sm = getattr(model, "exit_dropout", None) if dropout_exit else None
    head(x, model.linear, (model_exits(model) - 1), sm)


def model_exits(model):
    """Number of logits tensors the reference forward returns (4 multi-exit, 1 single-exit)."""
    if hasattr(model, "build_graph"):
        return int(model.n_exits)
    if getattr(model, "family", "") == "vgg":
        return 5 if getattr(model, "multi_exit", True) else 1
    return 4 if getattr(model, "multi_exit", True) else 1


def build_graph(model, device, dtype="f16"):
    g = GraphBuilder(device, dtype)
    fam = getattr(model, "family", None)
    if fam == "resnet":
        build_resnet_graph(model, g)
    elif fam == "vgg":
        from .models.vgg19.vgg19 import build_vgg_graph
        build_vgg_graph(model, g)
    elif hasattr(model, "build_graph"):
        model.build_graph(g)
    else:
        raise TypeError(f"{type(model).__name__} is not a bayesnn_fpga_amd model")
    return g


class CompiledGraph:
    """Host-only half of the engine: graph -> C descriptors -> bmi_create / bmi_plan / bmi_query.
    Touches no GPU API (weights only need to be addressable), so it also runs on a CPU-only box."""

    def __init__(self, model, device, max_batch, chunk_samples=None, dtype="f16"):
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self.n_exits = model_exits(model)
        self.out_dim = int(model.out_dim)
        self.dtype = dtype
        self.graph = build_graph(model, self.device, dtype)
        self.max_batch = int(max_batch)
        self.chunk_explicit = chunk_samples is not None
        if chunk_samples is None:
            chunk_samples = min(DEFAULT_CHUNK_SAMPLES_MAX, max(1, DEFAULT_CHUNK_IMAGES // self.max_batch))
        self.chunk_samples = int(chunk_samples)
        self._desc_keep = self._make_desc()
        self.handle = C.c_void_p()
        _lib.check(self.lib.bmi_create(C.byref(self._desc_keep[0]), C.byref(self.handle)), "bmi_create")
        ws = C.c_size_t()
        _lib.check(self.lib.bmi_plan(self.handle, self.max_batch, self.chunk_samples, C.byref(ws)), "bmi_plan")
        self.workspace_bytes = ws.value
        pm, sm, npo, nso = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.bmi_query(self.handle, C.byref(pm), C.byref(sm), C.byref(npo), C.byref(nso)), "bmi_query")
        self.prefix_macs, self.suffix_macs = pm.value, sm.value
        self.n_prefix_ops, self.n_suffix_ops = npo.value, nso.value
        # MACs that do not run in the MFMA conv kernels (bench.py's roofline accounting): the direct stem and the heads
        t = self.graph.tensors
        # whether the Philox seed reaches any kernel: Masksembles-only graphs (SA/utils.py: the masks are part of the state_dict) do not draw
        self.seed_matters = any((o.get("site") or {}).get("kind") in (_lib.SITE_ELEMENTWISE, _lib.SITE_CHANNEL) for o in self.graph.ops)
        self.stem_macs = sum(t[o["out"]][0] * t[o["out"]][1] * t[o["out"]][2] * 27 for o in self.graph.ops if o["kind"] == _lib.OP_STEM)
        self.head_macs = sum(t[o["in_"]][2] * self.out_dim for o in self.graph.ops if o["kind"] == _lib.OP_HEAD)
        self.dense_macs = sum(t[o["in_"]][2] * t[o["out"]][2] for o in self.graph.ops if o["kind"] == _lib.OP_DENSE)

    def _make_desc(self):
        g = self.graph
        tarr = (_lib.TensorDesc * len(g.tensors))(*[_lib.TensorDesc(*t) for t in g.tensors])
        oarr = (_lib.OpDesc * len(g.ops))()
        for i, op in enumerate(g.ops):
            d = oarr[i]
            d.kind, d.in_, d.out, d.residual = op["kind"], op["in_"], op["out"], op.get("residual", -1)
            d.ksize, d.stride, d.pad, d.relu = op.get("ksize", 0), op.get("stride", 0), op.get("pad", 0), op.get("relu", 0)
            d.in2 = op.get("in2", -1)
            d.site_pos = op.get("site_pos", _lib.SITE_POS_OUTER)
            for f in ("weight", "weight2", "scale", "bias", "bias_post"):
                t = op.get(f)
                setattr(d, f, t.data_ptr() if t is not None else None)
            s = op.get("site")
            if s:
                d.site = _lib.make_site(s["kind"], s["site_id"], s.get("p", 0.0), s.get("num_masks", 0),
                                        s["masks"].data_ptr() if "masks" in s else None)
            else:
                d.site = _lib.make_site()
        desc = _lib.ModelDesc(len(g.tensors), tarr, len(g.ops), oarr, self.n_exits, self.out_dim,
                              _lib.DTYPES[self.dtype])
        return (desc, tarr, oarr)

    def flops_per_batch(self, batch, T):
        """Executed-algorithmic FLOPs (prefix once + T x suffix), conv + linear only, 1 MAC = 2 FLOP."""
        return 2 * batch * (self.prefix_macs + T * self.suffix_macs)

    def conv_traffic_model(self, batch, T):
        """Algorithmic HBM bytes and launch count of the MFMA conv kernels for one batch x T samples: every conv reads
        its input (+ residual / shortcut input) and weights once and writes its output once, fp16.  A conv whose inputs
        are all deterministic runs once per batch (prefix), the others once per chunk of ``chunk_samples`` samples."""
        t, stoch = self.graph.tensors, set()
        chunks = -(-T // self.chunk_samples)
        total, launches = 0, 0
        for o in self.graph.ops:
            ins = [o["in_"]] + [o[k] for k in ("residual", "in2") if o.get(k, -1) is not None and o.get(k, -1) >= 0]
            det = not any(i in stoch for i in ins)
            if o["kind"] != _lib.OP_HEAD and (o.get("site") or not det):
                stoch.add(o["out"])
            if o["kind"] != _lib.OP_CONV:
                continue
            px = lambda i: t[i][0] * t[i][1] * t[i][2] * 2
            act = sum(px(i) for i in ins) + px(o["out"])
            wb = sum(o[k].numel() * 2 for k in ("weight", "weight2") if o.get(k) is not None)
            total += batch * act * (1 if det else T) + wb * (1 if det else chunks)
            launches += 1 if det else chunks
        return total, launches

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.bmi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MCDEngine(CompiledGraph):
    """One model compiled for one GPU.  All methods are asynchronous on the current torch stream
    (the stream handle is what the C ABI receives); results are device tensors."""

    def __init__(self, model, device, max_batch=256, chunk_samples=None, dtype="f16"):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("MCDEngine needs a HIP device (torch device type 'cuda' on ROCm); there is no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        super().__init__(model, device, max_batch, chunk_samples, dtype)
        self.workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=device)
        # the non-finite counter of bmi_finalize_checked: allocated HERE, never lazily — a first finalize() inside a hipGraph capture
        # (BatchesInFlight.predict_graphed) would otherwise allocate it from the graph's private pool and re-zero it on every replay
        self._nonfinite = torch.zeros(1, dtype=torch.int32, device=device)

    # ---- the path ------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_x(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.device == self.device):
            raise RuntimeError(f"input must live on {self.device}")
        if x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32):
            raise ValueError(f"expected float32 [B,3,32,32] like the reference's loaders, got {x.dtype} {tuple(x.shape)}")
        if x.shape[0] > self.max_batch:
            raise ValueError(f"batch {x.shape[0]} exceeds the engine's max_batch {self.max_batch}")
        return x.contiguous()

    def new_moments(self, batch):
        """Zeroed float64 accumulators S1 = sum p, S2 = sum p^2, SL = sum logit, each [E, B, C]."""
        return torch.zeros(3, self.n_exits, batch, self.out_dim, dtype=torch.float64, device=self.device)

    def accumulate(self, x, S, t_begin, t_count, seed=0, cnt0=0, image_offset=0):
        """Adds samples t_begin .. t_begin+t_count-1 of batch ``x`` into the moment buffer ``S``.
        ``image_offset``: ``x`` (and ``S``) are images image_offset.. of a larger batch — the masks are drawn at the images'
        indices in the whole batch (bmi_forward_mcd_images: one rank's share of a batch partitioned by images)."""
        x = self._check_x(x)
        B = x.shape[0]
        if tuple(S.shape) != (3, self.n_exits, B, self.out_dim) or S.dtype != torch.float64 or not S.is_contiguous():
            raise ValueError("moment buffer must be contiguous float64 [3, E, B, C]")
        with torch.cuda.device(self.device):
            rc = self.lib.bmi_forward_mcd_images(self.handle, x.data_ptr(), B, int(image_offset), int(t_begin), int(t_count),
                                                 int(seed) & 0xFFFFFFFFFFFFFFFF, int(cnt0), S[0].data_ptr(), S[1].data_ptr(),
                                                 S[2].data_ptr(), self.workspace.data_ptr(), self.workspace_bytes, self._stream())
        _lib.check(rc, "bmi_forward_mcd_images")
        return S

    def image_offset_ok(self, image_offset):
        """Whether a share of a batch that starts at image ``image_offset`` can be run by ``accumulate(..., image_offset=)``
        (bmi_image_offset_ok: host-only, every site's index offset must be a whole number of Philox calls)."""
        rc = self.lib.bmi_image_offset_ok(self.handle, int(image_offset))
        if rc not in (_lib.BMI_OK, -95):
            _lib.check(rc, "bmi_image_offset_ok")
        return rc == _lib.BMI_OK

    def finalize(self, S, t_total):
        """mean / var (ddof=0) / mean logit, float64 [E, B, C] each.  Also counts the non-finite sums into the engine's device counter
        (bmi_finalize_checked; no synchronisation here): ``check_finite()`` reads it when the results are read."""
        out = torch.empty_like(S)
        n = S[0].numel()
        with torch.cuda.device(self.device):
            rc = self.lib.bmi_finalize_checked(n, int(t_total), S[0].data_ptr(), S[1].data_ptr(), S[2].data_ptr(),
                                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), self._nonfinite.data_ptr(), self._stream())
        _lib.check(rc, "bmi_finalize_checked")
        return dict(mean=out[0], var=out[1], logit_mean=out[2])

    _nonfinite = None

    def nonfinite_count(self, reset=True):
        """Elements of the moment buffers finalized since the last reset whose sums were inf / NaN (a host read: synchronises)."""
        if self._nonfinite is None:
            return 0
        n = int(self._nonfinite.item())
        if reset and n:
            self._nonfinite.zero_()
        return n

    def check_finite(self):
        """Raises FloatingPointError if a finalize since the last check saw non-finite sums — on the 16-bit engines an activation past
        65 504 (fp16) turns into inf and then NaN in the softmax; the reference's fp32 path would have carried the value.  Callers that pull
        results to the host (FullAnalysis, evaluate, bench) call this right there."""
        n = self.nonfinite_count()
        if n:
            raise FloatingPointError(f"{n} non-finite moment sums on the {self.dtype!r} engine (overflow of a 16-bit activation?): "
                                     "use engine_dtype='f16x2' / 'bf16x3' (or 'auto', which checks this on the first batch)")

    def set_option(self, name, value):
        """A kernel-selection switch of THIS engine (bmi_engine_set_option): the engine was created with a copy of the process defaults
        (``_lib.set_option``) and keeps it whatever those become."""
        _lib.check(self.lib.bmi_engine_set_option(self.handle, name.encode(), int(value)), f"bmi_engine_set_option({name})")

    def predict(self, x, T, seed=0, t_begin=0, cnt0=0):
        S = self.new_moments(x.shape[0])
        self.accumulate(x, S, t_begin, T, seed, cnt0)
        return self.finalize(S, T)

    def predict_with_exit(self, x, T, threshold, seed=0, cnt0=0, first_exit=1):
        """Confidence-threshold early exiting on the device (bmi_forward_mcd_exit): an image leaves after the first exit
        e >= ``first_exit`` whose T-mean confidence exceeds ``threshold`` (the reference's ``confidence_exiting`` rule,
        SA/train/results_analyzer.py:606-630; its loop starts at exit 1) and the later stages only run for the images
        that are still active.  Returns dict(mean/var/logit_mean [E,B,C] — rows of exits an image never reached are
        meaningless —, exit_layer int32 [B], active_after [E] (host ints), best_preds [B,C] = mean[exit_layer[b], b])."""
        x = self._check_x(x)
        B = x.shape[0]
        if T > self.chunk_samples:
            raise ValueError(f"dynamic exiting needs all T={T} samples in one chunk (engine planned for {self.chunk_samples})")
        S = self.new_moments(B)
        exit_layer = torch.empty(B, dtype=torch.int32, device=self.device)
        active = (C.c_int32 * self.n_exits)()
        with torch.cuda.device(self.device):
            rc = self.lib.bmi_forward_mcd_exit(self.handle, x.data_ptr(), B, int(T), int(seed) & 0xFFFFFFFFFFFFFFFF, int(cnt0),
                                               float(threshold), int(first_exit), S[0].data_ptr(), S[1].data_ptr(), S[2].data_ptr(),
                                               exit_layer.data_ptr(), active, self.workspace.data_ptr(), self.workspace_bytes,
                                               self._stream())
        _lib.check(rc, "bmi_forward_mcd_exit")
        r = self.finalize(S, T)
        r["exit_layer"] = exit_layer
        r["active_after"] = [int(v) for v in active]
        r["best_preds"] = r["mean"][exit_layer.long(), torch.arange(B, device=self.device)]
        return r

    def forward_once(self, x, seed=0, t=0, cnt0=0):
        """One stochastic pass -> list of fp32 logits [B, C] per exit (the reference forward's return)."""
        S = self.new_moments(x.shape[0])
        self.accumulate(x, S, t, 1, seed, cnt0)
        return [S[2, e].float() for e in range(self.n_exits)]

    def forward_samples(self, x, T, seed=0, t_begin=0, cnt0=0, mask_stride=1, out=None):
        """Per-sample logits of the folded path (bmi_forward_mcd_samples): fp32 [T, E, B, C] — what T calls of the reference's
        ``model(x)`` return, from ONE pass of the engine over the batch (prefix once, the samples folded into the launches).
        ``mask_stride``: the Masksembles mask of sample i is (cnt0 + i * mask_stride) mod M (the reference's layers count forward
        calls: ``train/evaluate.py`` folds the T passes of batch k of an n-batch loader with cnt0 = cnt + k, mask_stride = n)."""
        x = self._check_x(x)
        B = x.shape[0]
        if out is None:
            out = torch.empty(T, self.n_exits, B, self.out_dim, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (T, self.n_exits, B, self.out_dim) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be contiguous float32 [T, E, B, C]")
        with torch.cuda.device(self.device):
            rc = self.lib.bmi_forward_mcd_samples(self.handle, x.data_ptr(), B, int(t_begin), int(T), int(seed) & 0xFFFFFFFFFFFFFFFF, int(cnt0),
                                                  int(mask_stride), out.data_ptr(), None, None, None, self.workspace.data_ptr(), self.workspace_bytes,
                                                  self._stream())
        _lib.check(rc, "bmi_forward_mcd_samples")
        return out

    def read_tensor(self, tensor_id, batch, samples=1):
        """A copy of graph tensor ``tensor_id`` as it sits in the workspace after a forward (bmi_tensor_info): fp32
        [samples * batch or batch, h, w, c].  For per-layer traces (tools/layer_trace.py): plan the engine under
        ``set_option("ws_no_reuse", 1)`` (and ``mask_lazy`` = 0, ``conv_pool`` = 0), else a later tensor may have taken the range."""
        off, eb, ps = C.c_int64(), C.c_int32(), C.c_int32()
        th, tw, tc = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.bmi_tensor_info(self.handle, int(tensor_id), C.byref(off), C.byref(eb), C.byref(ps), C.byref(th),
                                            C.byref(tw), C.byref(tc)), "bmi_tensor_info")
        n = batch * (samples if ps.value & 1 else 1)
        count = n * th.value * tw.value * tc.value
        raw = self.workspace[off.value:off.value + count * eb.value]
        if ps.value & 2:                   # the split engines' pair32 layout: per pixel, 32-channel blocks [32 heads | 32 tails]
            dt16 = torch.bfloat16 if self.dtype == "bf16x3" else torch.float16
            blocks = raw.view(dt16).view(n, th.value, tw.value, tc.value // 32, 2, 32).float()
            return (blocks[..., 0, :] + blocks[..., 1, :]).reshape(n, th.value, tw.value, tc.value).clone()
        dt = torch.float32 if eb.value == 4 else (torch.bfloat16 if self.dtype == "bf16" else torch.float16)
        return raw.view(dt).view(n, th.value, tw.value, tc.value).float().clone()

    # ---- measurement helpers ---------------------------------------------------------------------
    profiling = False

    def profile(self, enable):
        _lib.check(self.lib.bmi_profile_enable(self.handle, int(bool(enable))), "bmi_profile_enable")
        self.profiling = bool(enable)

    def profile_read(self):
        ms = (C.c_double * _lib.PROFILE_SLOTS)()
        n = (C.c_int64 * _lib.PROFILE_SLOTS)()
        _lib.check(self.lib.bmi_profile_read(self.handle, ms, n), "bmi_profile_read")
        out = {_lib.PROFILE_NAMES.get(i, str(i)): (ms[i], n[i]) for i in range(_lib.PROFILE_SLOTS) if n[i]}
        nf = _lib.CONV_FAMILIES
        fms, fn, ffl, fby = (C.c_double * nf)(), (C.c_int64 * nf)(), (C.c_double * nf)(), (C.c_double * nf)()
        _lib.check(self.lib.bmi_profile_conv_families(self.handle, fms, fn, ffl, fby), "bmi_profile_conv_families")
        names = _lib.CONV_FAMILY_KERNELS
        self.conv_families = {names[i]: dict(ms=fms[i], launches=fn[i], flops=ffl[i], bytes=fby[i]) for i in range(nf) if fn[i]}
        return out

    def profile_launches(self):
        """The launches behind the last profile_read(), in order: dicts with kind, family, out (tensor id), images, ms, flops, bytes."""
        cnt = C.c_int32(0)
        _lib.check(self.lib.bmi_profile_launches(self.handle, 0, C.byref(cnt), None, None, None, None, None, None, None), "bmi_profile_launches")
        n = cnt.value
        ki, fa, ou, im = ((C.c_int32 * n)() for _ in range(4))
        ms, fl, by = ((C.c_double * n)() for _ in range(3))
        _lib.check(self.lib.bmi_profile_launches(self.handle, n, C.byref(cnt), ki, fa, ou, im, ms, fl, by), "bmi_profile_launches")
        names = [k[:-len("_kernel")] for k in _lib.CONV_FAMILY_KERNELS]
        return [dict(kind=_lib.PROFILE_NAMES.get(ki[i], str(ki[i])), family=names[fa[i]] if 0 <= fa[i] < len(names) else None, out=ou[i], images=im[i],
                     ms=ms[i], flops=fl[i], bytes=by[i]) for i in range(n)]


class BatchesInFlight:
    """`n` independent engines of one model (own workspace, own device copy of the weights) on `n` streams: consecutive batches
    alternate between them, so the launch-bound once-per-batch prefix of batch k+1 (five ~50 us launches on B images) runs
    beside the sample-folded suffix of batch k instead of in front of it.  Nothing changes inside a batch — every per-sample
    value and every float64 moment sum is bit for bit the single-stream one (the 32-sample groups of an image are joined
    in group order) — only the order in which the GPU sees the launches of neighbouring batches changes.  Measured on
    one MI355X (tools/experiments/two_batches.py), 1 -> 2 batches in flight: VGG-11 T=30 0.368 -> 0.297 ms per batch, ResNet-18
    Masksembles T=8 2.52 -> 2.23 ms, ResNet-18 multi-exit at T=13 (one rank's share of eight) 3.77 -> 3.34 ms, at T=100 24.19 ->
    23.76 ms.

        pipe = BatchesInFlight(model, device, n=2, max_batch=250)
        for x in batches:
            out = pipe.submit(lambda eng: eng.predict(x, T, seed))     # asynchronous; device tensors
        pipe.synchronize()
    """

    def __init__(self, model, device, n=2, **engine_kwargs):
        if n < 1:
            raise ValueError("n >= 1 batches in flight")
        self._model = model
        self.engines = [MCDEngine(model, device, **engine_kwargs) for _ in range(n)]
        self.device = self.engines[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n)] if n > 1 else [None]
        self.last_stream = None
        self.last_engine = None
        self.k = 0

    def slot(self):
        return self.k % len(self.engines)

    use_graph = False      # step(): one hipGraph replay per batch step (tuned() sets it for launch-bound models)

    def close(self):
        """Destroys the engines and drops their workspaces, captured graphs and static buffers (a pipe that is being replaced by a larger
        one, or whose model's weights changed)."""
        self.synchronize()
        for attr in ("_graphs", "_gstreams"):
            if hasattr(self, attr):
                delattr(self, attr)
        for e in self.engines:
            e.close()
            e.workspace = None
            e.__dict__.pop("_step_S", None)
            e.__dict__.pop("_share_parts", None)
        self.engines = []

    @classmethod
    def tuned(cls, model, device, x, T, seed=0, cnt0=0, threshold_ms=1.5, allow_graph=True, group=None, **engine_kwargs):
        """The pipe a model should run with, decided by MEASUREMENT on its first batch: one engine is built, a batch step (x, T) is
        warmed up and timed with HIP events; a step under ``threshold_ms`` is launch-bound (VGG-11 at batch 250 x T = 30: ~20 launches of
        20-50 us on a ~20 us launch floor, 0.26 ms per step) and gets THREE batches in flight, each step ONE hipGraph replay
        (``predict_graphed``: 28 -> 38 M MCD-samples/s on VGG-11, round-3 measurement); anything longer (the ResNets at T = 100:
        21 ms) gets two eager engines, where a replay buys nothing and a third workspace costs memory.  ``pipe.step(x, T, seed)`` runs a
        batch either way; results are bit for bit the same in both modes (tests/test_gpu_model.py).  ``allow_graph=False``: a caller whose launch
        scalars change from batch to batch (FullAnalysis with MC-dropout sites: the batch index is part of the seed) takes the three batches in
        flight without the replay.  ``group``: the ranks of a process group time their SHARE of the step and decide together (MAX over ranks)."""
        pipe = cls(model, device, n=1, **engine_kwargs)
        from .sharding import _rank_world, accumulate_share
        rank, world = _rank_world(group) if group is not None else (0, 1)
        if world > 1:
            # a rank's SHARE of the step is what it will run; the group takes the slowest rank's figure so that every rank builds the same pipe
            import torch.distributed as dist
            S = pipe.engines[0].new_moments(x.shape[0])
            ms = pipe.measure_ms(lambda e: accumulate_share(e, x, S.zero_(), T, seed, cnt0, rank, world), warm=1, reps=2)
            on_dev = dist.get_backend(group) == "nccl"
            t = torch.tensor([ms], dtype=torch.float64, device=pipe.device if on_dev else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            ms = float(t.item())
        else:
            ms = pipe.measure_ms(lambda e: e.predict(x, T, seed, cnt0=cnt0), warm=1, reps=2)
        pipe.step_ms_measured = ms
        launch_bound = ms < threshold_ms
        pipe.grow(3 if launch_bound else 2)
        pipe.use_graph = bool(launch_bound and allow_graph)
        return pipe

    def measure_ms(self, fn, warm=2, reps=3):
        """Milliseconds per call of ``fn(engine 0)`` on the current stream (HIP events; ``warm`` untimed calls first)."""
        eng = self.engines[0]
        for _ in range(warm):
            fn(eng)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record(torch.cuda.current_stream(self.device))
        for _ in range(reps):
            fn(eng)
        ev[1].record(torch.cuda.current_stream(self.device))
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def grow(self, n):
        """``n`` batches in flight from now on: more engines of the same model (own workspace, own weights), one stream each."""
        if n < len(self.engines):
            raise ValueError("a pipe only grows")
        e0 = self.engines[0]
        kw = dict(max_batch=e0.max_batch, chunk_samples=e0.chunk_samples if e0.chunk_explicit else None, dtype=e0.dtype)
        self.engines += [MCDEngine(self._model, self.device, **kw) for _ in range(n - len(self.engines))]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n)] if n > 1 else [None]
        for attr in ("_graphs", "_gstreams"):
            if hasattr(self, attr):
                delattr(self, attr)
        return self

    def step(self, x, T, seed=0, cnt0=0, group=None, kind=None, shard=False):
        """One batch step on the next slot — a hipGraph replay (``use_graph``) or eager — with the work partitioned over ``group`` when
        one is given (``shard=True``: the default group).  Returns what ``MCDEngine.finalize`` returns (device tensors on ``last_stream``)."""
        if self.use_graph:
            return self.predict_graphed(x, T, seed, cnt0, group=group, kind=kind, shard=shard)
        from .sharding import accumulate_partitioned

        def run(e):
            S = e.__dict__.get("_step_S")
            if S is None or S.shape[2] != x.shape[0]:
                S = e.__dict__["_step_S"] = e.new_moments(x.shape[0])
            else:
                S.zero_()
            if group is not None or shard:
                accumulate_partitioned(e, x, S, T, seed, cnt0, group=group, kind=kind)
            else:
                e.accumulate(x, S, 0, T, seed, cnt0)
            return e.finalize(S, T)
        return self.submit(run, inputs=(x,))

    def submit(self, fn, inputs=()):
        """Runs fn(engine) on the next slot's stream (after everything already queued on the caller's current stream, which is
        where the inputs come from) and returns what it returns.  `inputs`: device tensors fn reads that the caller may drop
        before the batch has run (they are recorded on the stream so the caching allocator does not hand their memory out
        early).  The results are produced on `self.last_stream`: wait for it (stream.synchronize() / wait_stream) before
        reading them from another stream."""
        i = self.slot()
        self.k += 1
        st = self.streams[i]
        self.last_stream = st
        self.last_engine = self.engines[i]
        if st is None:
            return fn(self.engines[i])
        st.wait_stream(torch.cuda.current_stream(self.device))
        for t in inputs:
            t.record_stream(st)
        with torch.cuda.stream(st):
            return fn(self.engines[i])

    def predict_graphed(self, x, T, seed=0, cnt0=0, group=None, kind=None, shard=False, always_reduce=False):
        """``engine.predict(x, T, seed, cnt0=cnt0)`` of the next slot as ONE hipGraph launch (torch.cuda.CUDAGraph on ROCm).
        The library neither allocates nor synchronises inside bmi_forward_mcd / bmi_finalize, so the whole batch step — zero
        the moments, the once-per-batch prefix, every sample chunk of the suffix, finalize — is captured once per
        (slot, batch size, T, seed, cnt0) and replayed on the slot's static input buffer; a batch of another size (a loader's
        smaller last batch) is captured on first sight.  What it buys: the small-model configs are launch-bound — VGG-11 at
        batch 250 x T = 30 is 20 launches of 20-50 us on a ~20 us floor each — and a replay issues them back to back.
        Results are bit for bit the eager ones (tests/test_gpu_model.py).

        With a process group of more than one rank (``group``; ``shard=True`` takes the default group of an initialised ``torch.distributed``) the
        graph holds THIS RANK'S SHARE of the step (``sharding.accumulate_share``: its samples, or its images when T < ranks — one
        Masksembles mask of config 4 per GPU is exactly such a launch-bound step) and the all-reduce + finalize follow the replay
        eagerly on the slot's stream: a collective is never captured.

        The scalars of a launch (seed, first sample index, Masksembles counter) are baked into the captured kernel arguments:
        batches that must differ in them get a graph each — keep seed / cnt0 constant over a loader walk, or the cache (at most
        ``graph_cache_max`` graphs per slot, least recently used evicted with its static buffers) turns over.  Engine profiling
        (bmi_profile_enable records events) cannot be captured: refused.  A capture that fails runs the step eagerly instead.
        Returns the slot's STATIC output tensors: read them (after `last_stream`) before the slot comes round again, i.e. within
        the next len(engines) - 1 submissions."""
        from .sharding import _rank_world, accumulate_share
        # a share of the step only when the caller names the group (or shard=True for the default one): a data-parallel caller with a
        # different batch on each rank must not have its ranks' moments summed behind its back
        rank, world = _rank_world(group) if (group is not None or shard) else (0, 1)
        i = self.slot()
        eng = self.engines[i]
        if not hasattr(self, "_graphs"):
            from collections import OrderedDict
            self._graphs = [OrderedDict() for _ in self.engines]
            self._gstreams = [st if st is not None else torch.cuda.Stream(self.device) for st in self.streams]
        key = (tuple(x.shape), int(T), int(seed), int(cnt0), rank, world, kind, bool(always_reduce))
        cache = self._graphs[i]
        rec = cache.get(key)
        if rec is None and eng.profiling:          # (before the slot rotation advances: a refused call leaves the pipe as it was)
            raise RuntimeError("predict_graphed while engine profiling is on: the event records of bmi_profile_enable cannot be captured")
        self.k += 1
        st = self._gstreams[i]
        self.last_stream = st
        self.last_engine = eng
        cur = torch.cuda.current_stream(self.device)

        reduce = world > 1 or bool(always_reduce)        # (always_reduce: the collective in a group of ONE rank too — the 1-GPU RCCL probe)

        def eager():
            st.wait_stream(cur)
            x.record_stream(st)
            with torch.cuda.stream(st):
                S = eng.new_moments(x.shape[0])
                accumulate_share(eng, x, S, T, seed, cnt0, rank, world, kind)
                if reduce:
                    import torch.distributed as dist
                    dist.all_reduce(S, op=dist.ReduceOp.SUM, group=group)
                return eng.finalize(S, T)

        if rec is None:
            xs = torch.empty_like(x)
            S = eng.new_moments(x.shape[0])
            st.wait_stream(cur)
            with torch.cuda.stream(st):                          # warm-up on the capture stream (module load, first launches)
                xs.copy_(x)
                accumulate_share(eng, xs, S, T, seed, cnt0, rank, world, kind)
            st.synchronize()
            graph, out = torch.cuda.CUDAGraph(), {}
            try:
                with torch.cuda.graph(graph, stream=st):
                    S.zero_()
                    accumulate_share(eng, xs, S, T, seed, cnt0, rank, world, kind)
                    if not reduce:
                        out.update(eng.finalize(S, T))
            except Exception as exc:                              # (a launcher returned non-OK under capture, ...)
                import warnings
                warnings.warn(f"hipGraph capture of a batch step failed ({exc}); running eagerly")
                torch.cuda.synchronize(self.device)
                cache[key] = "eager"
                return eager()
            rec = cache[key] = (graph, xs, S, out)
            while len(cache) > self.graph_cache_max:
                cache.popitem(last=False)                          # least recently used: its graph and static buffers are freed
        if rec == "eager":
            return eager()
        cache.move_to_end(key)
        graph, xs, S, out = rec
        st.wait_stream(cur)
        x.record_stream(st)
        with torch.cuda.stream(st):
            xs.copy_(x, non_blocking=True)
            graph.replay()
            if reduce:
                import torch.distributed as dist
                dist.all_reduce(S, op=dist.ReduceOp.SUM, group=group)
                return eng.finalize(S, T)
        return out

    graph_cache_max = 8

    def synchronize(self):
        for st in list(self.streams) + list(getattr(self, "_gstreams", [])):
            if st is not None:
                st.synchronize()

