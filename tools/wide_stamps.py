#!/usr/bin/env python3
"""Reads the per-workgroup phase stamps of a -DBMI_WIDE_STAMPS build (conv_igemm_wide.hip) for one conv shape.
usage: python tools/wide_stamps.py D3 [images]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "D3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
SH = {"D3": (128, 256, 16, 3, 2, 1), "D4": (256, 512, 8, 3, 2, 1), "P4": (256, 512, 8, 1, 2, 0)}
cin, cout, H, k, s, p = SH[name]
ho = (H + 2 * p - k) // s + 1
lib = _lib.lib(); dev = "cuda:0"
g = torch.Generator().manual_seed(1)
x = torch.randn(n, H, H, cin, generator=g).half().to(dev); w = (torch.randn(cout, k, k, cin, generator=g) * 0.03).half().to(dev)
sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev)
out = torch.empty(n, ho, ho, cout, dtype=torch.float16, device=dev); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    _lib.check(lib.bmi_conv_igemm_fwd(x.data_ptr(), None, 1.0, w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None, out.data_ptr(), n, n, n, H, H, cin, cout, k, s, p, 1, None, 250, 0, 42, 0, st), "conv")
for _ in range(5): run()
torch.cuda.synchronize()
l = C.CDLL(_lib.LIB_PATH)
l.bmi_debug_wide_stamps_clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
buf = (C.c_ulonglong * (8192 * 8))(); l.bmi_debug_wide_stamps(buf, 8192 * 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
a = a[(a[:, 0] > 0) & (a[:, 3] > 0)]
t0 = a[:, 0].min()
nK = k * k * cin // 64
span = a[:, 3].max() - t0
print(name, "WGs stamped", len(a), "K-steps", nK, "span of the stamped WGs (ticks)", span, f"kernel elapsed {ms*1e3:.1f} us")
# slots 5/6: start / end on the 100 MHz real-time counter (comparable between workgroups)
rt = (a[:, 6] - a[:, 5]) / 100.0          # lifetime in us
span_us = (a[:, 6].max() - a[:, 5].min()) / 100.0
print(f"  real-time: launch span {span_us:.1f} us, median workgroup lifetime {np.median(rt):.2f} us "
      f"-> shader clock {np.median((a[:, 3] - a[:, 0]) / rt) / 1e3:.2f} GHz; CU occupancy by workgroups "
      f"sum(lifetime) / (256 CUs x span) = {rt.sum() / (256.0 * span_us):.3f}")
for nm, v in (("prologue", a[:, 1] - a[:, 0]), ("main", a[:, 2] - a[:, 1]), ("epilogue", a[:, 3] - a[:, 2]), ("lifetime", a[:, 3] - a[:, 0]),
              ("sum vmcnt wait (BMI_WIDE_STAMPS=2)", a[:, 4])):
    print(f"  {nm:18s} median {np.median(v):10.0f}  p10 {np.percentile(v,10):10.0f}  p90 {np.percentile(v,90):10.0f}   per K-step {np.median(v)/nK:8.1f}")
