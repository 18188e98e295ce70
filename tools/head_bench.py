#!/usr/bin/env python3
"""Times bmi_head_fused (pool + site + Linear + softmax + float64 moments) per class count / sample count.

    python tools/head_bench.py [--batch 250] [--K 512] [--hw 16] [--iters 20]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bayesnn_fpga_amd import _lib  # noqa: E402
import gpu_helpers as gh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--K", type=int, default=512)
    ap.add_argument("--hw", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cases", default="10:8,10:13,10:100,100:8,100:13,100:100")
    ap.add_argument("--site", default="elementwise", choices=["none", "elementwise", "masksemble"])
    a = ap.parse_args()
    lib, dev = _lib.lib(), "cuda:0"
    g = torch.Generator().manual_seed(1)
    B, K = a.batch, a.K
    for case in a.cases.split(","):
        out_dim, tc = (int(v) for v in case.split(":"))
        x = torch.randn(B * tc, a.hw, K, generator=g).half().to(dev)
        w = torch.zeros((out_dim + 31) // 32 * 32, K)
        w[:out_dim] = 0.05 * torch.randn(out_dim, K, generator=g)
        wd, bd = w.to(dev), torch.zeros(out_dim, device=dev)
        keep = []
        site = None
        if a.site == "elementwise":
            site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25)
        elif a.site == "masksemble":
            site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(torch.rand(8, K, generator=g) < 0.5).float().numpy())
        s = gh.site_struct(site, keep)
        S = torch.zeros(3, B, out_dim, dtype=torch.float64, device=dev)

        def run():
            _lib.check(lib.bmi_head_fused(gh.ptr(x), 0, B * tc, a.hw, K, gh.ptr(wd), gh.ptr(bd), out_dim, C.byref(s) if s is not None else None,
                                          None, B, 0, tc, 7, 0, gh.ptr(S[0]), gh.ptr(S[1]), gh.ptr(S[2]), gh.stream()), "bmi_head_fused")
        for _ in range(3):
            run()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.iters)
        ts.sort()
        in_gb = B * tc * a.hw * K * 2 / 1e9
        print(f"C={out_dim:4d} T={tc:4d} B={B} K={K} HW={a.hw}: median {ts[2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us   input {in_gb / ts[2] * 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
