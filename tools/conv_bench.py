#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels per GEMM shape class, through the
C ABI, timed with HIP events on the launch stream.  Used for kernel tuning and as the target of
rocprofv3 --pmc runs (few other kernels in the trace).

    python tools/conv_bench.py --images 1000 --iters 20 [--only S2,D3] [--check]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _lib  # noqa: E402

SHAPES = {  # Cin, Cout, H, k, stride, pad
    "S1": (64, 64, 32, 3, 1, 1), "D2": (64, 128, 32, 3, 2, 1), "P2": (64, 128, 32, 1, 2, 0),
    "S2": (128, 128, 16, 3, 1, 1), "D3": (128, 256, 16, 3, 2, 1), "P3": (128, 256, 16, 1, 2, 0),
    "S3": (256, 256, 8, 3, 1, 1), "D4": (256, 512, 8, 3, 2, 1), "P4": (256, 512, 8, 1, 2, 0),
    "S4": (512, 512, 4, 3, 1, 1),
    # the stride-2 launches of the ResNet-18 multi-exit suffix as the engine issues them (layerN[0].conv1 + the exit head's first
    # conv as one pair launch = twice the channels)
    "D2p": (64, 256, 32, 3, 2, 1), "D3p": (128, 512, 16, 3, 2, 1), "D4p": (256, 1024, 8, 3, 2, 1),
    # Bottleneck 1x1 convs of ResNet-50 (HBM-bound): expand (with residual in the network) and reduce
    "E2": (128, 512, 16, 1, 1, 0), "E3": (256, 1024, 8, 1, 1, 0), "E4": (512, 2048, 4, 1, 1, 0),
    "Q1": (256, 128, 32, 1, 1, 0), "R2": (512, 128, 16, 1, 1, 0), "R3": (1024, 256, 8, 1, 1, 0), "R4": (2048, 512, 4, 1, 1, 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=1, help="timed rounds per variant (interleaved)")
    ap.add_argument("--ab", default="", help="comma list of option sets for a same-process A/B, e.g. "
                                             "mfma_shape_patch=32+mfma_shape_wide=32,mfma_shape_patch=16+mfma_shape_wide=16")
    ap.add_argument("--site", action="store_true", help="fuse an elementwise MC-dropout site into the epilogue")
    ap.add_argument("--nores", action="store_true")
    ap.add_argument("--in-mod", type=int, default=0, help="the input tensor holds this many images (a deterministic tensor under sample folding)")
    ap.add_argument("--noscale", action="store_true")
    ap.add_argument("--zero-input", action="store_true", help="all-zero activations: the same instruction stream at the lowest operand-toggling power")
    ap.add_argument("--sparse-input", action="store_true", help="post-ReLU, 25 %% dropped activations (as inside the network) instead of N(0,1)")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = "cuda:0"
    names = [s for s in a.only.split(",") if s] or [k for k in SHAPES if k[0] not in "EQR" and not k.endswith("p")]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    total_t = total_f = 0.0
    for name in names:
        cin, cout, H, k, s, p = SHAPES[name]
        n = a.images
        ho = (H + 2 * p - k) // s + 1
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n, H, H, cin, generator=g)
        if a.zero_input:
            x = torch.zeros_like(x)
        if a.sparse_input:
            x = torch.relu(x) * (torch.rand(n, H, H, cin, generator=g) > 0.25)
        x = x.to(torch.float16).to(dev)
        w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.float16).to(dev)
        scale = (0.5 + torch.rand(cout, generator=g)).to(dev)
        bias = (0.1 * torch.randn(cout, generator=g)).to(dev)
        res = torch.randn(n, ho, ho, cout, generator=g).to(torch.float16).to(dev)
        out = torch.empty(n, ho, ho, cout, dtype=torch.float16, device=dev)
        site = _lib.make_site(_lib.SITE_ELEMENTWISE, 2, 0.25) if a.site else None

        def run():
            rc = lib.bmi_conv_igemm_fwd(x.data_ptr(), None, 1.0, w.data_ptr(), None if a.noscale else scale.data_ptr(),
                                        None if a.noscale else bias.data_ptr(), None if a.nores else res.data_ptr(),
                                        out.data_ptr(), n, a.in_mod or n, n, H, H, cin, cout, k, s, p, 1,
                                        C.byref(site) if site is not None else None, 250, 0, 42, 0, st)
            _lib.check(rc, "bmi_conv_igemm_fwd")
        flops = 2.0 * n * ho * ho * cout * k * k * cin
        variants = [v for v in a.ab.split(",") if v] or [""]      # e.g. mfma_shape_patch=32,mfma_shape_patch=16
        times = {v: [] for v in variants}

        def select(v):
            for kv in (v.split("+") if v else []):
                nm, _, val = kv.partition("=")
                _lib.set_option(nm, int(val))
        for v in variants:
            select(v)
            for _ in range(3):
                run()
        for _ in range(a.rounds):                       # interleaved rounds in ONE process: variance is correlated
            for v in variants:
                select(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.iters)
        for v in variants:
            t = sorted(times[v])
            ms = t[len(t) // 2]
            nbytes = 2.0 * (n * H * H * cin + n * ho * ho * cout * (1 if a.nores else 2) + cout * k * k * cin)
            print(f"{name} {v or 'default':28s}: median {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  {nbytes / ms / 1e6:6.0f} GB/s   min {t[0] * 1e3:8.1f} us "
                  f"{flops / t[0] / 1e9:7.1f} TFLOP/s   (M={n * ho * ho}, N={cout}, K={k * k * cin})", flush=True)
        total_t += sorted(times[variants[0]])[len(times[variants[0]]) // 2]
        total_f += flops
    print(f"all: {total_f / total_t / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
