#!/usr/bin/env python3
"""Soak test of conv3x3_s2's masked-input form: many repeats of the same launch (persistent walk, several tiles per workgroup) with a
copy kernel hammering HBM on a second stream; every output must equal the first one bit for bit."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bayesnn_fpga_amd import _lib
from oracle import philox
from tests import gpu_helpers as gh
from tests.test_gpu_kernels import _conv_inputs

lib = _lib.lib()
# argv[2] = "half": the 128-channel tiles (256 -> 128, ResNet-50's lazy reader) instead of the 64 -> 128 + 128 pair shape
HALF = len(sys.argv) > 2 and sys.argv[2] == "half"
B, tc, cin, cout, H, t0, seed, p = (25, 40, 256, 128, 32, 3, (7 << 32) + 5, 0.25) if HALF else (25, 40, 64, 256, 32, 3, (7 << 32) + 5, 0.25)
N = B * tc
x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, B, 321, False)
x = torch.relu(x)
keep = []
s = gh.site_struct(dict(kind=_lib.SITE_ELEMENTWISE, site_id=1, p=p), keep)
bits = torch.zeros(N * H * H * cin // 8, dtype=torch.uint8, device=gh.DEV)
_lib.check(lib.bmi_mask_bits(gh.ptr(bits), N, H * H, cin, C.byref(s), B, t0, seed, gh.stream()), "bmi_mask_bits")
xs = (x.float() * float(philox.drop_scale(p))).to(torch.float16)
junk_a = torch.empty(256 << 20, dtype=torch.uint8, device=gh.DEV)
junk_b = torch.empty_like(junk_a)
side = torch.cuda.Stream()
first, bad = None, 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for i in range(reps):
    with torch.cuda.stream(side):
        junk_b.copy_(junk_a, non_blocking=True)
    out = gh.run_conv(xs, w, scale, bias, None, True, 2, 1, N, B, 1, batch=B, in_bits=bits)
    if first is None:
        first = out.clone()
    elif not torch.equal(out, first):
        bad += 1
torch.cuda.synchronize()
print(f"{reps} launches, {bad} differed from the first")
sys.exit(1 if bad else 0)
