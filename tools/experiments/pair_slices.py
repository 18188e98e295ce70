#!/usr/bin/env python3
"""Probe: does slicing a Bottleneck seam (expand conv + residual of block k, reduce conv of block k+1) by images keep the wide
tensor in the Infinity Cache between its write and its read?   E(all) R(all)   vs   E(s0) R(s0) E(s1) R(s1) ...   (same kernels,
same bits), optionally with the two launch kinds on two streams.
    python tools/experiments/pair_slices.py --layer 2 --images 16000
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _lib  # noqa: E402

LAYERS = {2: (128, 512, 16), 3: (256, 1024, 8), 4: (512, 2048, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=2)
    ap.add_argument("--images", type=int, default=16000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--slices", default="0,2000,1000,500,250")
    a = ap.parse_args()
    lib = _lib.lib()
    dev = "cuda:0"
    cm, cw, H = LAYERS[a.layer]
    n = a.images
    g = torch.Generator().manual_seed(1)
    m = torch.randn(n, H, H, cm, generator=g).to(torch.float16).to(dev)
    res = torch.randn(n, H, H, cw, generator=g).to(torch.float16).to(dev)
    w3 = (torch.randn(cw, 1, 1, cm, generator=g) * (2.0 / cm) ** 0.5).to(torch.float16).to(dev)
    w1 = (torch.randn(cm, 1, 1, cw, generator=g) * (2.0 / cw) ** 0.5).to(torch.float16).to(dev)
    s3, b3 = (0.5 + torch.rand(cw, generator=g)).to(dev), (0.1 * torch.randn(cw, generator=g)).to(dev)
    s1, b1 = (0.5 + torch.rand(cm, generator=g)).to(dev), (0.1 * torch.randn(cm, generator=g)).to(dev)
    y = torch.empty(n, H, H, cw, dtype=torch.float16, device=dev)
    z = torch.empty(n, H, H, cm, dtype=torch.float16, device=dev)
    px = H * H
    esz = 2

    def E(lo, cnt, st):
        _lib.check(lib.bmi_conv_igemm_fwd(m.data_ptr() + lo * px * cm * esz, None, 1.0, w3.data_ptr(), s3.data_ptr(), b3.data_ptr(),
                                          res.data_ptr() + lo * px * cw * esz, y.data_ptr() + lo * px * cw * esz, cnt, cnt, cnt, H, H, cm, cw,
                                          1, 1, 0, 1, None, 250, 0, 42, 0, st), "E")

    def R(lo, cnt, st):
        _lib.check(lib.bmi_conv_igemm_fwd(y.data_ptr() + lo * px * cw * esz, None, 1.0, w1.data_ptr(), s1.data_ptr(), b1.data_ptr(), None,
                                          z.data_ptr() + lo * px * cm * esz, cnt, cnt, cnt, H, H, cw, cm, 1, 1, 0, 1, None, 250, 0, 42, 0, st), "R")

    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    st0, st1 = C.c_void_p(main_s.cuda_stream), C.c_void_p(side.cuda_stream)
    def seam():
        _lib.check(lib.bmi_conv1x1_seam_fwd(m.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), res.data_ptr(), y.data_ptr(), w1.data_ptr(),
                                            s1.data_ptr(), b1.data_ptr(), z.data_ptr(), n, H, H, cm, cw, cm, 1, st0), "seam")
    E(0, n, st0)
    R(0, n, st0)
    torch.cuda.synchronize()
    y0, z0 = y.clone(), z.clone()
    y.zero_()
    z.zero_()
    seam()
    torch.cuda.synchronize()
    same = torch.equal(y, y0) and torch.equal(z, z0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        seam()
    e1.record()
    torch.cuda.synchronize()
    print(f"layer{a.layer} n={n} one conv1x1_seam launch        {e0.elapsed_time(e1) / a.iters:7.3f} ms  same_bits={same}", flush=True)
    ref = None
    for spec in a.slices.split(","):
        S = int(spec)
        for two in ((False,) if S == 0 else (False, True)):
            def step():
                if S == 0:
                    E(0, n, st0)
                    R(0, n, st0)
                    return
                evs = []
                for lo in range(0, n, S):
                    cnt = min(S, n - lo)
                    E(lo, cnt, st0)
                    if two:      # R(slice) on the side stream behind E(slice); E(next slice) runs beside it
                        ev = torch.cuda.Event()
                        ev.record(main_s)
                        side.wait_event(ev)
                        R(lo, cnt, st1)
                    else:
                        R(lo, cnt, st0)
                if two:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    main_s.wait_event(ev)
            step()
            torch.cuda.synchronize()
            if ref is None:
                ref = (y.clone(), z.clone())
            same = torch.equal(y, ref[0]) and torch.equal(z, ref[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            print(f"layer{a.layer} n={n} slice={S or 'all':>5} two_streams={int(two)}  {e0.elapsed_time(e1) / a.iters:7.3f} ms  same_bits={same}", flush=True)


if __name__ == "__main__":
    main()
