#!/usr/bin/env python3
"""Bitwise comparison of two bmi_set_option settings on single convs:  ab_bitwise.py wide_direct_w=0 wide_direct_w=1"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _lib  # noqa: E402

lib, dev = _lib.lib(), "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(3)
variants = sys.argv[1:3]
for cin, cout, H, k, s, p, n in [(64, 256, 32, 3, 2, 1, 300), (128, 256, 16, 3, 2, 1, 1100), (256, 512, 8, 3, 2, 1, 4200), (256, 512, 8, 1, 2, 0, 5000), (128, 512, 16, 1, 1, 0, 2001), (64, 128, 32, 1, 2, 0, 777), (256, 1024, 8, 1, 1, 0, 3003),
                                 (256, 512, 8, 1, 1, 0, 3000), (128, 256, 16, 3, 2, 1, 4001)]:
    x = torch.randn(n, H, H, cin, generator=g).half().to(dev)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).half().to(dev)
    sc, bi = (0.5 + torch.rand(cout, generator=g)).to(dev), (0.1 * torch.randn(cout, generator=g)).to(dev)
    ho = (H + 2 * p - k) // s + 1
    res = torch.randn(n, ho, ho, cout, generator=g).half().to(dev)
    for use_res, use_site in [(0, 0), (1, 1)]:
        site = _lib.make_site(_lib.SITE_ELEMENTWISE, 2, 0.25) if use_site else None
        outs = []
        for v in variants:
            for kv in v.split("+"):
                nm, _, val = kv.partition("=")
                _lib.set_option(nm, int(val))
            out = torch.full((n, ho, ho, cout), float("nan"), dtype=torch.float16, device=dev)
            _lib.check(lib.bmi_conv_igemm_fwd(x.data_ptr(), None, 1.0, w.data_ptr(), sc.data_ptr(), bi.data_ptr(),
                                              res.data_ptr() if use_res else None, out.data_ptr(), n, n, n, H, H, cin, cout, k, s, p, 1,
                                              C.byref(site) if site is not None else None, 250, 0, 42, 0, st), "conv")
            torch.cuda.synchronize()
            outs.append(out)
        a, b = outs
        ne = (a.view(torch.int16) != b.view(torch.int16))
        d = (a.float() - b.float()).abs()
        print(f"{(cin, cout, H, k, s, n)} res={use_res} site={use_site}: {int(ne.sum())} of {a.numel()} differ, max {float(d.nan_to_num(9e9).max()):.3g}, "
              f"nan {int(torch.isnan(b).sum())}", flush=True)
