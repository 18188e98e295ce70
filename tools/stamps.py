#!/usr/bin/env python3
"""Reads the per-workgroup phase stamps of a -DBMI_PATCH_STAMPS build for one conv shape."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _lib
sys.argv += [] 
name = sys.argv[1] if len(sys.argv) > 1 else "S2"
NORES = "--nores" in sys.argv          # plain epilogue (BN + ReLU only) instead of the general one
SH = {"S2": (128, 128, 16), "S3": (256, 256, 8), "S4": (512, 512, 4)}
cin, cout, H = SH[name]
lib = _lib.lib(); n = 1000; dev = "cuda:0"
g = torch.Generator().manual_seed(1)
x = torch.randn(n, H, H, cin, generator=g).half().to(dev); w = (torch.randn(cout, 3, 3, cin, generator=g) * 0.03).half().to(dev)
sc = torch.ones(cout, device=dev); bi = torch.zeros(cout, device=dev); res = torch.randn(n, H, H, cout, generator=g).half().to(dev)
out = torch.empty(n, H, H, cout, dtype=torch.float16, device=dev); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    _lib.check(lib.bmi_conv_igemm_fwd(x.data_ptr(), None, 1.0, w.data_ptr(), sc.data_ptr(), bi.data_ptr(), None if NORES else res.data_ptr(), out.data_ptr(), n, n, n, H, H, cin, cout, 3, 1, 1, 1, None, 250, 0, 42, 0, st), "conv")
for _ in range(5): run()
torch.cuda.synchronize()
l = C.CDLL(_lib.LIB_PATH)
l.bmi_debug_stamps_clear(); run(); torch.cuda.synchronize()
buf = (C.c_ulonglong * (8192 * 8))(); l.bmi_debug_stamps(buf, 8192 * 8)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print(name, "WGs", len(a), "kernel span (cycles @100MHz?)", (a[:, 3].max() - t0))
for nm, v in (("prologue", a[:, 1] - a[:, 0]), ("main", a[:, 2] - a[:, 1]), ("epilogue", a[:, 3] - a[:, 2]), ("lifetime", a[:, 3] - a[:, 0]),
              ("slot4 (vmcnt wait)", a[:, 4]), ("slot5 (wait+barrier)", a[:, 5])):
    print(f"  {nm:18s} median {np.median(v):10.0f}  p10 {np.percentile(v,10):10.0f}  p90 {np.percentile(v,90):10.0f}")
starts = np.sort(a[:, 0] - t0)
print("  start times: first wave of WGs ends at", starts[min(511, len(starts)-1)], " last start", starts[-1])
