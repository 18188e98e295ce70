#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; never on the GPU box).  The
reference's Python is imported from where it lies (PYTHONPATH), never copied; only
input/output vectors are written.  Recipe: SURVEY.md Appendix D.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

What is patched in the reference: ONLY the RNG primitive — ``torch.nn.functional.dropout``
is replaced by the shared Philox mask (oracle/philox.py), which the reference's
``MCDropout.forward`` resolves at call time (SA/models/resnet18/resnet18.py:210).
``torchvision`` / ``KDEpy`` (absent here) are stubbed so that ``train.results_analyzer``
imports; nothing from the stubs is executed on the path.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Software_Artifact/software"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle import philox  # noqa: E402
from oracle.layers import MCContext, philox_dropout  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def _stub_modules():
    tv = types.ModuleType("torchvision")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.datasets"] = tv.datasets
    sys.modules["torchvision.transforms"] = tv.transforms
    kde = types.ModuleType("KDEpy")
    kde.FFTKDE = None
    sys.modules["KDEpy"] = kde


_stub_modules()
import utils as ref_utils  # noqa: E402  (reference SA/utils.py)
from models.resnet18.resnet18 import ResNet18MCEarlyExit, ResNet18MC, ResNet18EarlyExit  # noqa: E402
import models as ref_models  # noqa: E402
from train.results_analyzer import FullAnalysis  # noqa: E402
from train.loss.base_classes import _MultiExitAccuracy  # noqa: E402

CTX = MCContext()
_orig_dropout = torch.nn.functional.dropout


def _patched_dropout(x, p=0.5, training=True, inplace=False):
    assert training
    return philox_dropout(CTX, x, p)


class philox_patch:
    def __enter__(self):
        torch.nn.functional.dropout = _patched_dropout

    def __exit__(self, *a):
        torch.nn.functional.dropout = _orig_dropout


def state_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def ref_passes(model, x, T, seed, reset_cnt=True):
    """T forwards of the reference model under the Philox patch -> float32 [T,E,B,C]."""
    model.eval()
    if reset_cnt:
        for m in model.modules():
            if hasattr(m, "cnt"):
                m.cnt = 0
    outs = []
    with torch.no_grad(), philox_patch():
        for t in range(T):
            CTX.begin_forward(seed, t)
            outs.append(np.stack([o.numpy() for o in model(x)]))
    return np.stack(outs)


RESNET_CASES = {
    # name: (ctor kwargs, B, T)
    "exit_only": (dict(dropout_exit=True, dropout=None, dropout_p=0.25, out_dim=10), 4, 6),
    "block_exit": (dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 4, 6),
    "block_noexit": (dict(dropout_exit=False, dropout="block", dropout_p=0.5, out_dim=10), 3, 4),
    "layer_exit": (dict(dropout_exit=True, dropout="layer", dropout_p=0.125, out_dim=10), 4, 4),
    "mask4_block_exit": (dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0,
                              out_dim=10), 4, 10),
    "mask8_exit_c100": (dict(dropout_exit=True, dropout=None, mask_type="mask", num_masks=8, mask_scale=4.0,
                             out_dim=100), 2, 8),
    # p values that draw 16 and 8 bits per element (the cases above draw 2 and 4): oracle/philox.py site_bits
    "block_exit_p02": (dict(dropout_exit=True, dropout="block", dropout_p=0.2, out_dim=10), 5, 4),
    "layer_exit_p256": (dict(dropout_exit=True, dropout="layer", dropout_p=3.0 / 256.0, out_dim=10), 3, 3),
}


def gen_resnet():
    for name, (kw, B, T) in RESNET_CASES.items():
        torch.manual_seed(0)
        np.random.seed(0)
        model = ResNet18MCEarlyExit(**kw)
        init_sum = state_checksum(model.state_dict())
        synthetic_weights_(model, 0)
        x = synthetic_images(B, seed=1234)
        seed = 42
        logits = ref_passes(model, x, T, seed)
        # the reference's own T-loop (FullAnalysis._get_output) on the same stream
        fa = FullAnalysis.__new__(FullAnalysis)
        fa.model, fa.mc_dropout, fa.mc_passes = model, True, T
        fa.outputs = list(range(model.n_exits))
        fa.device = torch.device("cpu")
        for m in model.modules():
            if hasattr(m, "cnt"):
                m.cnt = 0
        state = {"t": 0}

        def pre(mod, inp):
            CTX.begin_forward(seed, state["t"])
            state["t"] += 1
        hnd = model.register_forward_pre_hook(pre)
        with torch.no_grad(), philox_patch():
            out, out_sm, out_sm_np, ens_out, ens_sm = fa._get_output(x)
        hnd.remove()
        masks = {k: v.numpy().astype(np.uint8) for k, v in model.state_dict().items() if k.endswith(".masks")}
        np.savez_compressed(
            os.path.join(OUT, f"resnet18_{name}.npz"),
            kwargs=repr(kw), B=B, T=T, seed=seed, init_checksum=init_sum,
            weights_checksum=state_checksum(model.state_dict()),
            logits=logits.astype(np.float32),
            go_output=np.stack([o.numpy() for o in out]),
            go_output_sm=np.stack([o.numpy() for o in out_sm]),
            go_output_sm_np=np.asarray(out_sm_np),
            go_ensemble_output=np.stack([o.numpy() for o in ens_out]),
            go_ensemble_output_sm=np.stack([o.numpy() for o in ens_sm]),
            **{"mask__" + k: v for k, v in masks.items()},
        )
        print("resnet18", name, logits.shape, init_sum[:12])

    # single-exit ResNet18MC and the deterministic early-exit net
    torch.manual_seed(0)
    np.random.seed(0)
    m = ResNet18MC(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    init_sum = state_checksum(m.state_dict())
    synthetic_weights_(m, 0)
    x = synthetic_images(3, seed=1234)
    np.savez_compressed(os.path.join(OUT, "resnet18mc_block_exit.npz"), B=3, T=4, seed=7, init_checksum=init_sum,
                        logits=ref_passes(m, x, 4, 7).astype(np.float32))
    torch.manual_seed(0)
    m = ResNet18EarlyExit(out_dim=10)
    init_sum = state_checksum(m.state_dict())
    synthetic_weights_(m, 0)
    m.eval()
    with torch.no_grad():
        lg = np.stack([o.numpy() for o in m(x)])
    np.savez_compressed(os.path.join(OUT, "resnet18_early_exit.npz"), B=3, init_checksum=init_sum,
                        logits=lg.astype(np.float32))
    # factory dispatch (SA/models/model_loader.py:8-24)
    hp = dict(call="ResNet18", resnet_type="mc_early_exit", load_model=None, out_dim=10, image_size=32,
              dropout="block", dropout_exit=True, dropout_p=0.25, n_exits=4, mask_type="mc", num_masks=4,
              mask_scale=4.0)
    torch.manual_seed(0)
    net = ref_models.get_network(hp)
    np.savez_compressed(os.path.join(OUT, "factory.npz"), hp=repr(hp), cls=type(net).__name__,
                        keys=np.array(sorted(net.state_dict().keys())),
                        init_checksum=state_checksum(net.state_dict()))
    print("factory", type(net).__name__)


VGG_CASES = {
    "exit_mc": (dict(dropout_exit=True, dropout=None, dropout_p=0.25, out_dim=100, n_exits=5), 2, 4),
    "exit_mask4": (dict(dropout_exit=True, dropout=None, mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10,
                        n_exits=5), 2, 6),
}


def gen_vgg():
    from models.vgg19.vgg19 import VGG19MCEarlyExit, VGG19MC
    for name, (kw, B, T) in VGG_CASES.items():
        torch.manual_seed(0)
        np.random.seed(0)
        model = VGG19MCEarlyExit(**kw)
        init_sum = state_checksum(model.state_dict())
        synthetic_weights_(model, 0)
        x = synthetic_images(B, seed=1234)
        logits = ref_passes(model, x, T, 42)
        np.savez_compressed(os.path.join(OUT, f"vgg19_{name}.npz"), kwargs=repr(kw), B=B, T=T, seed=42,
                            init_checksum=init_sum, weights_checksum=state_checksum(model.state_dict()),
                            logits=logits.astype(np.float32))
        print("vgg19", name, logits.shape)
    torch.manual_seed(0)
    m = VGG19MC(dropout_exit=True, dropout_p=0.5, out_dim=10)
    init_sum = state_checksum(m.state_dict())
    synthetic_weights_(m, 0)
    x = synthetic_images(3, seed=1234)
    np.savez_compressed(os.path.join(OUT, "vgg19mc_exit.npz"), B=3, T=3, seed=7, init_checksum=init_sum,
                        logits=ref_passes(m, x, 3, 7).astype(np.float32))
    # the broken insertion modes: both raise AttributeError at construction
    broken = []
    for cls in (VGG19MC, VGG19MCEarlyExit):
        for mode in ("block", "layer"):
            try:
                cls(dropout=mode, dropout_exit=True, out_dim=10)
                broken.append("ok")
            except Exception as e:      # noqa: BLE001
                broken.append(type(e).__name__)
    np.savez_compressed(os.path.join(OUT, "vgg19_broken_modes.npz"), errors=np.array(broken))
    print("vgg broken modes", broken)


def gen_masksembles():
    np.random.seed(3)
    m2 = ref_utils.Masksembles2D(16, 4, 2.0).eval()
    m1 = ref_utils.Masksembles1D(32, 4, 2.0).eval()
    g = torch.Generator().manual_seed(5)
    x2 = torch.randn(3, 16, 5, 5, generator=g)
    x1 = torch.randn(3, 32, generator=g)
    y2 = np.stack([m2(x2).numpy() for _ in range(8)])
    y1 = np.stack([m1(x1).numpy() for _ in range(8)])
    props = []
    for (c, n, s) in [(512, 4, 4.0), (512, 8, 4.0), (64, 4, 4.0), (128, 4, 4.0), (256, 4, 4.0), (64, 4, 3.0),
                      (512, 4, 6.0)]:
        np.random.seed(11)
        mk = ref_utils.generation_wrapper(c, n, s)
        props.append((c, n, s, mk.shape[0], mk.shape[1], int(mk.sum(1)[0]), int((mk.sum(1) == mk.sum(1)[0]).all()),
                      hashlib.sha256(mk.astype(np.uint8).tobytes()).hexdigest()))
    np.savez_compressed(os.path.join(OUT, "masksembles.npz"),
                        masks2=m2.masks.numpy(), masks1=m1.masks.numpy(), x2=x2.numpy(), x1=x1.numpy(), y2=y2, y1=y1,
                        props=np.array(props, dtype=object), allow_pickle=True)
    print("masksembles ok")


def gen_evaluate_masksembles():
    """The reference's OWN evaluate() (SA/train/evaluate.py:8-22: T outer passes over the loader, validate_model_acc per pass,
    _MultiExitAccuracy._metrics per batch) on a Masksembles model — deterministic, nothing patched: its layers count their forward
    calls, so pass i of batch k of an n-batch loader is call i * n + k and sees mask (i * n + k) mod M.  Pins the folded route's
    mask_stride (bmi_forward_mcd_samples: the T passes of batch k in one call, masks (cnt + k + i n) mod M)."""
    from train.evaluate import evaluate as ref_evaluate
    kw = dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)
    B, nb, T = 4, 3, 5                                   # 3 batches, M = 4: stride 3, the walk visits masks 0,3,2,1,0 / 1,0,3,2,1 / 2,1,0,3,2
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(ResNet18MCEarlyExit(**kw), 0).eval()
    x, y = synthetic_images(B * nb, seed=3), synthetic_labels(B * nb, 10, seed=4)
    loader = [(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    loss = _MultiExitAccuracy(4, acc_tops=(1, 5))
    cnt0 = 2                                             # the counters do not start at 0 (a model that has been called before)
    for m in model.modules():
        if hasattr(m, "cnt") and hasattr(m, "masks"):
            m.cnt = cnt0
    with torch.no_grad():
        avg = ref_evaluate(loss, loader, model, -1, "golden", T, create_log=False)
    cnt_after = [int(m.cnt) for m in model.modules() if hasattr(m, "cnt") and hasattr(m, "masks")]
    assert len(set(cnt_after)) == 1
    np.savez_compressed(os.path.join(OUT, "evaluate_masksembles.npz"), kwargs=str(kw), B=B, nb=nb, T=T, cnt0=cnt0, cnt_after=cnt_after[0],
                        metric_names=np.array(loss.metric_names), averaged=np.array([float(v) for v in avg]))
    print("evaluate_masksembles", [round(float(v), 4) for v in avg], cnt_after[0])


def gen_metrics():
    rng = np.random.RandomState(17)
    N, C = 2000, 10
    p = rng.dirichlet(np.ones(C) * 0.3, size=N)
    y = rng.randint(0, C, size=N)
    # make ~70 % of the argmaxes correct so the set is not degenerate
    am = p.argmax(1)
    flip = rng.rand(N) < 0.7
    y = np.where(flip, am, y)
    onehot = np.eye(C)[y]
    fa = FullAnalysis.__new__(FullAnalysis)
    ece = float(fa.ece_hist_binary(p, onehot).item())
    mse = np.mean(np.sum((p - onehot) ** 2, 1))                       # results_analyzer.py:498
    pc = np.clip(p, 1e-256, 1 - 1e-256)
    nll = -np.sum(onehot * np.log(pc)) / N                              # :500-501
    acc = np.sum((np.argmax(pc, 1) - np.array([np.where(r == 1)[0][0] for r in onehot])) == 0) / N   # :502
    # _metrics accuracy vector
    g = torch.Generator().manual_seed(23)
    logits_list = [torch.randn(64, C, generator=g) * 2 for _ in range(4)]
    yy = torch.randint(0, C, (64,), generator=g)
    acc_vec4 = [float(v) for v in _MultiExitAccuracy(4, acc_tops=(1, 5))._metrics(logits_list, yy)]
    acc_vec1 = [float(v) for v in _MultiExitAccuracy(1, acc_tops=(1, 5))._metrics(logits_list, yy)]
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), p=p, onehot=onehot, ece_hist=ece, nll=nll, mse=mse, acc=acc,
                        logits=np.stack([l.numpy() for l in logits_list]), y=yy.numpy(),
                        acc_vec4=np.array(acc_vec4), acc_vec1=np.array(acc_vec1))
    print("metrics", ece, nll, mse, acc)


def gen_confidence_exiting():
    """Reference confidence-threshold exiting + FLOP model on a fixed random prediction set."""
    rng = np.random.RandomState(5)
    E, N, C = 4, 300, 10
    logits = rng.randn(E, N, C) * np.array([1.0, 1.5, 2.0, 3.0])[:, None, None]
    p = torch.softmax(torch.from_numpy(logits), -1).numpy()
    y = rng.randint(0, C, size=N)
    y = np.where(rng.rand(N) < 0.6, p[-1].argmax(1), y)
    onehot = np.eye(C)[y]
    fa = FullAnalysis.__new__(FullAnalysis)
    fa.model_type = "resnet18"
    fa.get_flops_per_module()
    fa.ece_eval_binary = lambda bp, lab: (bp.copy(), 0.0, 0.0, float(np.mean(bp.argmax(1) == lab.argmax(1))))
    out = dict(p=p, onehot=onehot, thresholds=np.array([0.1, 0.5, 0.8, 0.95, 0.999]), baseline=fa.baseline_flops)
    for k, th in enumerate(out["thresholds"]):
        for diff in (False, True):
            acc, best, _ = fa.confidence_exiting(float(th), p, onehot, diff=diff)
            out[f"best_{k}_{int(diff)}"] = best
            out[f"acc_{k}_{int(diff)}"] = acc
        for eo in (True, False):
            fa.exit_only = eo
            out[f"flops_{k}_{int(eo)}"] = fa.flop_saver(float(th), p, onehot, mc_passes=10)
            out[f"ensflops_{k}_{int(eo)}"] = fa.flop_saver_ensembled(float(th), p, onehot, mc_passes=10)
    out["std_exit"] = np.array([[fa.get_flops_standard_exit(l, 10, ens) for l in range(4)] for ens in (False, True)])
    np.savez_compressed(os.path.join(OUT, "confidence_exiting.npz"), **out)
    print("confidence exiting ok")


def gen_philox():
    """Mask bits of the shared convention for a few (seed, site, t, p, shape) — pins the
    layout rule (NHWC-linear element order) independently of the model fixtures."""
    cases = []
    for (seed, site, t, p, shape) in [(42, 0, 0, 0.25, (2, 8, 3, 3)), (42, 3, 7, 0.5, (3, 12)),
                                       ((1 << 40) + 5, 1, 99, 0.125, (1, 4, 2, 5)), (0, 0, 0, 0.0, (1, 8)),
                                       (9, 2, 1, 1.0, (1, 8))]:
        cases.append(dict(seed=seed, site=site, t=t, p=p, shape=shape,
                          mask=philox.elementwise_mask(shape, seed, site, t, p).astype(np.uint8)))
    np.savez_compressed(os.path.join(OUT, "philox_masks.npz"), cases=np.array(cases, dtype=object), allow_pickle=True)


def converter_cnn():
    """The small sequential CNN of the converter fixtures (shared with tests/helpers.py by construction order)."""
    from torch import nn
    return nn.Sequential(
        nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(64, 128, 3, padding=1, bias=False), nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2, 2)),
        nn.Sequential(nn.Conv2d(256, 256, 3, padding=1), nn.BatchNorm2d(256), nn.ReLU()),
        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(256, 10))


def _import_ref_converter():
    """The reference's Dropouts.py / nn2bnn.py.  nn2bnn.py imports a file that is not in the repository
    (``test.ThreeLayerNet``, :5): an empty module of that name is registered so that the import statement passes;
    nothing of it is used."""
    conv_dir = "/root/reference/Hardware_Artifact/converter/pytorch"
    sys.path.insert(0, conv_dir)
    saved_test = sys.modules.get("test"), sys.modules.get("test.ThreeLayerNet")
    t = types.ModuleType("test")
    tl = types.ModuleType("test.ThreeLayerNet")
    tl.ThreeLayerNet = None
    t.ThreeLayerNet = tl
    sys.modules["test"], sys.modules["test.ThreeLayerNet"] = t, tl
    import Dropouts as ref_dropouts
    import nn2bnn as ref_nn2bnn
    for k, v in zip(("test", "test.ThreeLayerNet"), saved_test):
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    sys.path.remove(conv_dir)
    return ref_dropouts, ref_nn2bnn


def _run_reference_converter(net, fixture, B, T, seed, p, returns_list, extra=None):
    """The reference's nn2bnn._convert_model (Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45) applied to ``net``, T passes of
    the converted model's own forward on the seeded synthetic batch, written as tests/golden/<fixture>.npz.  Patched: only
    F.dropout / F.dropout2d (the Bernoulli source), as for the other fixtures."""
    ref_dropouts, ref_nn2bnn = _import_ref_converter()

    def d2(x, p=0.5, training=True, inplace=False):
        assert training
        return philox_dropout(CTX, x, p, channelwise=True)

    init = state_checksum(net.state_dict())
    synthetic_weights_(net, 0)
    wsum = state_checksum(net.state_dict())
    model = ref_nn2bnn._convert_model(net, p)
    classes = [type(m).__name__ for m in model.modules() if isinstance(m, ref_dropouts._DropoutBase)]
    x = synthetic_images(B, seed=1234)
    model.eval()
    outs = []
    sites_per_pass = 0
    orig = ref_dropouts.F.dropout, ref_dropouts.F.dropout2d
    ref_dropouts.F.dropout, ref_dropouts.F.dropout2d = _patched_dropout, d2
    try:
        with torch.no_grad():
            for tt in range(T):
                CTX.begin_forward(seed, tt)
                out = model(x)
                if returns_list:
                    assert isinstance(out, list)
                    outs.append(np.stack([o.numpy() for o in out]))
                else:
                    outs.append(out.numpy()[None])
                sites_per_pass = CTX.site
    finally:
        ref_dropouts.F.dropout, ref_dropouts.F.dropout2d = orig
    logits = np.stack(outs)                      # [T, E, B, C]
    fields = dict(logits=logits, B=B, T=T, seed=seed, p=p, init_checksum=init, weights_checksum=wsum,
                  wrapper_classes=np.array(classes), keys=np.array(list(model.state_dict().keys())))
    fields.update(extra(ref_dropouts) if extra else dict(sites_per_pass=sites_per_pass))
    np.savez(os.path.join(OUT, fixture + ".npz"), **fields)
    print(fixture, logits.shape, len(classes), "wrappers,", sites_per_pass, "sites per pass", float(np.abs(logits).max()))


def gen_converter_resnet():
    """... on the reference's own ResNet18Base (SA/models/resnet18/resnet18.py:189-204): every Conv2d — the stem, both convs of
    every BasicBlock, the 1x1 shortcut convs, the (unused) exit-head convs — becomes BayesianDropout2D, every Linear
    BayesianDropout."""
    from models.resnet18.resnet18 import ResNet18Base
    torch.manual_seed(0)
    _run_reference_converter(ResNet18Base(n_exits=1, out_dim=10), "converter_resnet18base", 3, 4, 77, 0.25, True)


def gen_converter_vgg():
    """... on the reference's VGG19 (SA/models/vgg19/vgg19.py:186-192; VGG.forward :107-119): 16 wrapped convs (site before the
    BatchNorm), 5 wrapped MaxPool2d (elementwise), the wrapped classifier (logits).  The converter also wraps the second
    references to the same layers in ``non_sequentialized_blocks`` (unused by the forward)."""
    from models.vgg19.vgg19 import VGG19
    torch.manual_seed(0)
    _run_reference_converter(VGG19(n_exits=1, out_dim=10), "converter_vgg19", 3, 4, 55, 0.25, True)


def gen_converter_multi_exit():
    """... on the reference's own MULTI-EXIT classes (round 4): ResNet18EarlyExit (SA/models/resnet18/resnet18.py:182-186, forward
    :144-180: 4 logits; the exit-head convs are live here, 27 sites per pass) and VGG19EarlyExit (SA/models/vgg19/vgg19.py:256-324:
    5 logits; the wrapped MaxPool2d at the end of a block feeds both the next block and the exit in front of it)."""
    from models.resnet18.resnet18 import ResNet18EarlyExit
    from models.vgg19.vgg19 import VGG19EarlyExit
    torch.manual_seed(0)
    _run_reference_converter(ResNet18EarlyExit(n_exits=4, out_dim=10), "converter_resnet18ee", 3, 4, 91, 0.25, True)
    torch.manual_seed(0)
    _run_reference_converter(VGG19EarlyExit(n_exits=5, out_dim=10), "converter_vgg19ee", 3, 4, 92, 0.25, True)


def gen_converter_custom():
    """... on a small net with a HAND-WRITTEN forward (tests/helpers.py:converter_custom_net: a residual add, functional ReLU and
    pooling, ``.view``, two outputs): the reference's _convert_model takes any nn.Module; the package compiles this one through
    torch.fx (converter/pytorch/fx_frontend.py)."""
    from tests.helpers import converter_custom_net
    torch.manual_seed(0)
    _run_reference_converter(converter_custom_net(), "converter_custom", 5, 6, 4321, 0.25, True)


def gen_converter():
    """... on a small sequential CNN (tests/helpers.py:converter_cnn), plus the reference's ValueError text for p outside [0, 1]."""
    def extra(ref_dropouts):
        err = ""
        try:
            ref_dropouts.BayesianDropout(torch.nn.Linear(2, 2), p=1.5)
        except ValueError as e:
            err = str(e)
        return dict(bad_p_error=err)
    torch.manual_seed(0)
    _run_reference_converter(converter_cnn(), "converter_cnn", 5, 6, 1234, 0.25, False, extra)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "converter":
        gen_converter()
        gen_converter_resnet()
        gen_converter_vgg()
        gen_converter_multi_exit()
        gen_converter_custom()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "evaluate_masksembles":
        gen_evaluate_masksembles()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "converter_custom":
        gen_converter_custom()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "converter_multi_exit":
        gen_converter_multi_exit()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "converter_resnet":
        gen_converter_resnet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "converter_vgg":
        gen_converter_vgg()
        sys.exit(0)
    gen_converter()
    gen_converter_resnet()
    gen_converter_vgg()
    gen_converter_multi_exit()
    gen_converter_custom()
    gen_philox()
    gen_masksembles()
    gen_evaluate_masksembles()
    gen_metrics()
    gen_confidence_exiting()
    gen_resnet()
    gen_vgg()
