#!/usr/bin/env python3
"""Where the folded evaluate() route spends a batch: the engine pass alone (resident input), + the batched metric op, + the host-to-device copy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bayesnn_fpga_amd.engine import BatchesInFlight
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy
wl = bench.WORKLOADS["resnet18_me"]; dev = torch.device("cuda:0")
model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
B, T, n = 250, int(sys.argv[1]) if len(sys.argv) > 1 else 10, 40
xh, yh = synthetic_images(B, seed=1), synthetic_labels(B, 10, seed=2)
x, y = xh.to(dev), yh.to(dev)
loss = MultiExitAccuracy(4)
for nfl in (1, 2):
    pipe = BatchesInFlight(model, dev, n=nfl, max_batch=B)
    def t(fn, what):
        for _ in range(3): fn()
        pipe.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        pipe.synchronize(); torch.cuda.synchronize()
        print(f"in_flight={nfl} {what:40} {(time.perf_counter() - t0) / n * 1e3:.3f} ms per batch", flush=True)
    t(lambda: pipe.submit(lambda e: e.forward_samples(x, T, seed=1)), "forward_samples (resident x)")
    t(lambda: pipe.submit(lambda e: e.predict(x, T, seed=1)), "predict (moments, resident x)")
    t(lambda: pipe.submit(lambda e: loss._metrics_passes(e.forward_samples(x, T, seed=1), y)), "+ _metrics_passes")
    t(lambda: pipe.submit(lambda e: loss._metrics_passes(e.forward_samples(xh.to(dev, non_blocking=True), T, seed=1), yh.to(dev, non_blocking=True))), "+ H2D of a pageable batch")
    lg = pipe.engines[0].forward_samples(x, T, seed=1)
    t(lambda: loss._metrics_passes(lg, y), "_metrics_passes alone")
