#!/usr/bin/env python3
"""Which launch differs between the lazy and the materialised first site?  Two engines of one model (ws_no_reuse: every tensor keeps its
workspace range), one planned with mask_lazy = 1, one with 0; every op's output compared bit for bit.
    python tools/experiments/lazy_vs_plain_trace.py [--model resnet50_me] [--batch 250] [--T 2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from bayesnn_fpga_amd import _lib  # noqa: E402
from bayesnn_fpga_amd.engine import MCDEngine  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images  # noqa: E402
from layer_trace import KIND, make  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50_me")
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--T", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = make(a.model).to(dev).eval()
    x = synthetic_images(a.batch, seed=1234).to(dev)
    _lib.set_option("ws_no_reuse", 1)
    res, eng = {}, {}
    for lazy in (1, 0):
        _lib.set_option("mask_lazy", lazy)
        eng[lazy] = MCDEngine(model, dev, max_batch=a.batch, chunk_samples=a.T, dtype="f16")
        res[lazy] = eng[lazy].predict(x, a.T, seed=3)
        torch.cuda.synchronize()
    ops = eng[1].graph.ops
    for i, o in enumerate(ops):
        if o["kind"] == _lib.OP_HEAD:
            continue
        try:
            ta = eng[1].read_tensor(o["out"], a.batch, a.T)
            tb = eng[0].read_tensor(o["out"], a.batch, a.T)
        except _lib.BmiError as e:
            print(f"{i:3d} {KIND[o['kind']]:7} (not readable: {e})")
            continue
        nd = int((ta != tb).sum())
        print(f"{i:3d} {KIND[o['kind']]:7} out {o['out']:3d} {str(tuple(eng[1].graph.tensors[o['out']])):>16}  differing elements {nd:9d} of {ta.numel()}  max|diff| {float((ta.double() - tb.double()).abs().max()):.3e}")
    for k in ("mean", "var"):
        print(k, float((res[1][k] - res[0][k]).abs().max()))


if __name__ == "__main__":
    main()
