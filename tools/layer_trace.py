"""Per-layer error trace: the fp16 (or bf16) engine against the EXACT engine (dtype "f32", csrc/conv_exact.hip) on the same model,
inputs and masks — every activation tensor of the graph, read back from the two workspaces (bmi_tensor_info).

What it answers (round-3 review, What's weak #1): when a 16-bit run differs from the fp32 reference by 6e-2 on a logit, is that
rounding growing through the layers or one layer misbehaving?  A misbehaving layer shows as a jump of the relative error at its
row; rounding shows as a slow drift of ~1e-3 x sqrt(depth) of the activation scale.

    python tools/layer_trace.py --model converter_resnet18base [--dtype f16] [--T 2] [--batch 4]
    models: converter_resnet18base | converter_vgg19 | converter_cnn | resnet18_block_exit | vgg11 | resnet50_me

Both engines are planned under ws_no_reuse (every tensor keeps its own workspace range), mask_lazy = 0 and conv_pool = 0 (every
tensor is materialised), BMI_FUSE_SHORTCUT=0 (the same graph in both engines: tensor ids line up).
"""
import argparse
import os
import sys

os.environ["BMI_FUSE_SHORTCUT"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bayesnn_fpga_amd import _lib  # noqa: E402
from bayesnn_fpga_amd.engine import MCDEngine  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402

KIND = {_lib.OP_STEM: "stem", _lib.OP_CONV: "conv", _lib.OP_MASK: "mask", _lib.OP_HEAD: "head", _lib.OP_MAXPOOL: "maxpool", _lib.OP_DENSE: "dense"}


def make(name):
    from bayesnn_fpga_amd.converter.pytorch import MCDropout
    from bayesnn_fpga_amd.models import extra as bx
    from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18Base, ResNet18MCEarlyExit
    from bayesnn_fpga_amd.models.vgg19.vgg19 import VGG19
    torch.manual_seed(0)
    np.random.seed(0)
    if name == "converter_resnet18base":
        return MCDropout(synthetic_weights_(ResNet18Base(n_exits=1, out_dim=10), 0), nSamples=4, p=0.25)
    if name == "converter_vgg19":
        return MCDropout(synthetic_weights_(VGG19(n_exits=1, out_dim=10), 0), nSamples=4, p=0.25)
    if name == "converter_cnn":
        from tests.helpers import converter_cnn
        return MCDropout(synthetic_weights_(converter_cnn(), 0), nSamples=4, p=0.25)
    if name == "resnet18_block_exit":
        return synthetic_weights_(ResNet18MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 0)
    if name == "vgg11":
        return synthetic_weights_(bx.VGG11MC(num_bayes_layer=3, dropout_p=0.25, out_dim=10), 0)
    if name == "resnet50_me":
        return synthetic_weights_(bx.ResNet50MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 0)
    raise SystemExit(f"unknown model {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="converter_resnet18base")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--T", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = make(a.model).to(dev).eval()
    x = synthetic_images(a.batch, seed=1234).to(dev)
    for k, v in (("ws_no_reuse", 1), ("mask_lazy", 0), ("conv_pool", 0)):
        _lib.set_option(k, v)
    eng = {dt: MCDEngine(model, dev, max_batch=a.batch, chunk_samples=a.T, dtype=dt) for dt in (a.dtype, "f32")}
    out = {}
    for dt, e in eng.items():
        out[dt] = e.predict(x, a.T, seed=a.seed)
    torch.cuda.synchronize()
    ops = eng["f32"].graph.ops
    assert [o["kind"] for o in ops] == [o["kind"] for o in eng[a.dtype].graph.ops], "the two graphs differ"
    print(f"# layer trace: {a.model}, {a.dtype} engine vs exact (f32) engine, batch {a.batch}, T {a.T}, seed {a.seed}")
    print(f"# {'op':>3} {'kind':7} {'out':>4} {'shape (h,w,c)':>15} {'site':>5} | {'max|exact|':>10} {'rms exact':>10} {'max|diff|':>10} {'rms diff':>10} {'rms diff / rms exact':>21}")
    worst = (0.0, -1)
    for i, o in enumerate(ops):
        if o["kind"] == _lib.OP_HEAD:
            continue
        tid = o["out"]
        try:
            ta = eng[a.dtype].read_tensor(tid, a.batch, a.T).double()
            tb = eng["f32"].read_tensor(tid, a.batch, a.T).double()
        except _lib.BmiError:
            continue
        d = (ta - tb)
        rms_e, rms_d = float(tb.pow(2).mean().sqrt()), float(d.pow(2).mean().sqrt())
        rel = rms_d / max(rms_e, 1e-30)
        site = o.get("site")
        st = "-" if not site else {_lib.SITE_ELEMENTWISE: "elt", _lib.SITE_CHANNEL: "chan", _lib.SITE_MASKSEMBLE: "mask"}[site["kind"]]
        h, w, c = eng["f32"].graph.tensors[tid]
        print(f"  {i:3d} {KIND[o['kind']]:7} {tid:4d} {str((h, w, c)):>15} {st:>5} | {float(tb.abs().max()):10.4f} {rms_e:10.4f} {float(d.abs().max()):10.3e} {rms_d:10.3e} {rel:21.3e}")
        if rel > worst[0]:
            worst = (rel, i)
    for k in ("mean", "var", "logit_mean"):
        print(f"# {k}: max|{a.dtype} - exact| = {float((out[a.dtype][k] - out['f32'][k]).abs().max()):.3e}   (max|exact| = {float(out['f32'][k].abs().max()):.3f})")
    print(f"# largest relative rms error: {worst[0]:.3e} at op {worst[1]}")


if __name__ == "__main__":
    main()
