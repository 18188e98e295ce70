#!/usr/bin/env python3
"""Precision probe for Winograd F(2x2, 3x3) on this path (CPU emulation): every stride-1 3x3 conv of the ResNet-18
multi-exit suffix is evaluated (a) directly on fp16-rounded operands with fp32 accumulation (what the HIP kernels do)
and (b) through the Winograd transforms with the transformed input and the transformed weights rounded to fp16 before
the element-wise products (what an MFMA Winograd kernel would feed the matrix cores), fp32 accumulation and fp32 output
transform.  Reports the error of the MC-dropout predictive mean against the fp32 oracle for both.

    python tools/experiments/winograd_error.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402
from oracle import mcd  # noqa: E402
from oracle import resnet18 as oresnet  # noqa: E402

G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
MODE = {"v": "fp32"}


def h(x):
    return x.half().float()


def winograd_conv(x, w):
    """x [N,C,H,W] (H, W even), w [K,C,3,3], pad 1, stride 1.  fp16-rounded U and V, fp32 accumulate."""
    N, C, H, W = x.shape
    K = w.shape[0]
    U = h(torch.einsum("ij,kcjl,ml->kcim", G, w, G))                        # [K,C,4,4]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # [N,C,H/2,W/2,4,4]
    V = h(torch.einsum("ij,nctujl,ml->nctuim", BT, tiles, BT))              # [N,C,th,tw,4,4]
    M = torch.einsum("kcim,nctuim->nktuim", U, V)                           # fp32 accumulate over c
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, M, AT)                     # [N,K,th,tw,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


_orig = F.conv2d


def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    s = stride if isinstance(stride, int) else stride[0]
    p = padding if isinstance(padding, int) else padding[0]
    if MODE["v"] == "fp32":
        return _orig(x, w, b, stride, padding, *a, **k)
    xh, wh = h(x), h(w)
    if MODE["v"] == "wino" and w.shape[2] == 3 and s == 1 and p == 1 and x.shape[1] >= 64 and x.shape[2] % 2 == 0:
        y = winograd_conv(xh, wh)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return _orig(xh, wh, b, stride, padding, *a, **k)


def main():
    torch.manual_seed(0)
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m = synthetic_weights_(oresnet.ResNet18MCEarlyExit(**kw), 0)
    x = synthetic_images(16, seed=1234)
    F.conv2d = conv2d
    torch.nn.functional.conv2d = conv2d
    out = {}
    for mode in ("fp32", "fp16", "wino"):
        MODE["v"] = mode
        out[mode] = mcd.mcd_predict(m, x, 4, 42)
    for mode in ("fp16", "wino"):
        dm = np.abs(out[mode]["mean"] - out["fp32"]["mean"]).max(axis=(1, 2))
        dl = np.abs(out[mode]["logits"] - out["fp32"]["logits"]).max()
        print(f"{mode}: max |mean - fp32| per exit {np.array2string(dm, precision=5)}   max |logit - fp32| {dl:.4f}")


if __name__ == "__main__":
    main()
