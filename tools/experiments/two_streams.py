#!/usr/bin/env python3
"""Does running two half-size sample chunks on two HIP streams beat one full chunk on one stream?  (The non-MFMA kernels
of one chunk — mask, heads — could overlap the other chunk's convs.)  Two engines (own workspaces), T/2 samples each."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd.engine import MCDEngine  # noqa: E402
from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = synthetic_weights_(ResNet18MCEarlyExit(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 0).to(dev).eval()
B, T = 250, 100
x = synthetic_images(B, seed=1).to(dev)
e1 = MCDEngine(m, dev, max_batch=B)
ea, eb = MCDEngine(m, dev, max_batch=B, chunk_samples=T // 2), MCDEngine(m, dev, max_batch=B, chunk_samples=T // 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    S = e1.new_moments(B)
    e1.accumulate(x, S, 0, T, 5)
    return S


def two():
    Sa, Sb = ea.new_moments(B), eb.new_moments(B)
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ea.accumulate(x, Sa, 0, T // 2, 5)
    with torch.cuda.stream(sb):
        eb.accumulate(x, Sb, T // 2, T // 2, 5)
    cur.wait_stream(sa); cur.wait_stream(sb)
    return Sa + Sb


for name, fn in (("one stream, T=100", one), ("two streams, 2 x T=50", two), ("one stream, T=100", one), ("two streams, 2 x T=50", two)):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt * 1e3:.2f} ms  -> {B * T / dt:.0f} samples/s")
print("max |difference| of the moments:", float((one() - two()).abs().max()))
