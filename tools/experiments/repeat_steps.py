#!/usr/bin/env python3
"""Determinism stress of the whole path under two batches in flight: N steps of the headline config, every predictive mean /
variance compared bit for bit with the first step's."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesnn_fpga_amd.engine import BatchesInFlight  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "resnet18_me"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
torch.manual_seed(0)
np.random.seed(0)
model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
B, T = wl[3], wl[4]
pipe = BatchesInFlight(model, dev, n=2, max_batch=B)
x = synthetic_images(B, seed=1234).to(dev)
outs = [pipe.submit(lambda e: e.predict(x, T, seed=42)) for _ in range(steps)]
torch.cuda.synchronize()
bad = sum(1 for o in outs[1:] if not (torch.equal(o["mean"], outs[0]["mean"]) and torch.equal(o["var"], outs[0]["var"])))
ref = outs[len(outs) // 2]
bad_mid = sum(1 for o in outs if not torch.equal(o["mean"], ref["mean"]))
dmax = max(float((o["mean"] - outs[0]["mean"]).abs().max()) for o in outs[1:])
dmid = max(float((o["mean"] - ref["mean"]).abs().max()) for o in outs)
vd = [float((o["var"] - ref["var"]).abs().max()) for o in outs]
print("steps whose var differs from the middle step:", [(i, f"{d:.3g}") for i, d in enumerate(vd) if d > 0][:12])
print(f"{steps} steps, 2 in flight: {bad} differ from the first (max |d mean| {dmax:.3g}); {bad_mid} differ from step {len(outs) // 2} (max {dmid:.3g})")
sys.exit(1 if dmax > 1e-12 else 0)
