#!/usr/bin/env python3
"""Two batches in flight: consecutive bench steps alternate between two engines (own workspace, own moment buffers) on two
streams, so the launch-bound once-per-batch prefix of step k+1 can run beside the suffix of step k.

    python tools/experiments/two_batches.py --workload vgg11 [--T 13]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesnn_fpga_amd.engine import MCDEngine  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vgg11", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
    B, T = wl[3], a.T or wl[4]
    x = synthetic_images(B, seed=1234).to(dev)
    for n_eng in (1, 2, 3):
        engs = [MCDEngine(model, dev, max_batch=B) for _ in range(n_eng)]
        Ss = [e.new_moments(B) for e in engs]
        streams = [torch.cuda.Stream() for _ in range(n_eng)]
        outs = [None] * n_eng

        def step(k):
            i = k % n_eng
            with torch.cuda.stream(streams[i]):
                Ss[i].zero_()
                engs[i].accumulate(x, Ss[i], 0, T, 42)
                outs[i] = engs[i].finalize(Ss[i], T)
        for k in range(6):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ref = outs[0]["mean"].clone() if n_eng == 1 else ref0
        if n_eng == 1:
            ref0 = ref
        same = all(torch.equal(o["mean"], ref0) for o in outs)
        print(f"{a.workload} T={T}: {n_eng} batch(es) in flight: {dt / a.steps * 1e3:8.3f} ms/step  {B * T * a.steps / dt:12.0f} samples/s   "
              f"results equal to the single-stream run: {same}", flush=True)
        del engs, Ss


if __name__ == "__main__":
    main()
