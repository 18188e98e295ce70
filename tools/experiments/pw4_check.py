#!/usr/bin/env python3
"""conv3x3_pw4 (four waves, software-pipelined) against conv3x3_pw: same K order -> the same bits; plain and residual + site epilogues."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesnn_fpga_amd import _lib
from tests import gpu_helpers as gh
from tests.test_gpu_kernels import _conv_inputs

bad = 0
for cin, cout, H, n in [(256, 256, 8, 21), (256, 256, 8, 1030), (512, 512, 4, 37), (512, 512, 4, 4111), (64, 256, 8, 9), (128, 512, 4, 100)]:
    for use_res, use_site in [(False, False), (True, False), (True, True)]:
        x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, n, 500 + n, False)
        res = torch.randn(n, H, H, cout, generator=g).to(torch.float16).to(gh.DEV) if use_res else None
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25) if use_site else None
        outs = {}
        for mode in (2, 4):
            _lib.set_option("conv_pw", mode)
            outs[mode] = gh.run_conv(x, w, scale, bias, res, True, 1, 1, n, n, n, site=site, batch=n, t0=1, seed=3)
        _lib.set_option("conv_pw", 1)
        eq = torch.equal(outs[2].view(torch.int16), outs[4].view(torch.int16))
        ref = gh.conv_ref(x, w, scale, bias, res, True, 1, 1, n, n, n)
        if site is not None:
            ref = ref * gh.folded_site_mask(site, n, cout, H, H, 1, 1, 3)
        err = (outs[4].float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        print(f"{cin}->{cout} {H}x{H} n={n} res={use_res} site={use_site}: pw4 == pw: {eq}, max|pw4 - ref| = {err:.2e}")
        bad += (not eq) or err > 2e-2
print("BAD" if bad else "OK")
sys.exit(1 if bad else 0)
