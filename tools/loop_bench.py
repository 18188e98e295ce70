#!/usr/bin/env python3
"""End-to-end loop throughput: the reference's unit of work is the loader walk of FullAnalysis.sdn_get_detailed_results
(SA/train/results_analyzer.py:113-177: 40 batches of 250 images, host -> device per batch, T stochastic passes, host-side
collation of predictions / labels / per-instance trackers).  This times the package's FullAnalysis mirror over a synthetic
10 000-image HOST-resident loader (torch DataLoader, batch 250, the reference's test-loader settings) — H2D copies, the two
batches in flight, the .cpu() of the moments and the tracker updates all inside the clock — next to the device-only figure of
the same batches (inputs resident in HBM, no collation: what bench.py reports).

    python tools/loop_bench.py [--workload resnet18_me] [--images 10000] [--batch 250] [--T 100] [--repeats 3]

Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesnn_fpga_amd.engine import BatchesInFlight  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_  # noqa: E402
from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis  # noqa: E402


def evaluate_route(a, wl, model, loader):
    """train.evaluate over the host loader: `passes` outer passes, each model(X) a single-sample launch-bound step (25 launches on 250
    images), metrics per batch.  Two figures: the mirror as shipped (the metric vectors stay on the device, one host synchronisation per
    pass over the loader) and with the reference's per-batch host synchronisation (loss_f.metrics: nine .cpu() per batch)."""
    from bayesnn_fpga_amd.engine import model_exits
    from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate

    n_exits = model_exits(model)
    out = {}
    # folded (the default since round 5): one walk over the loader, the T passes of a batch as ONE engine pass (MCDEngine.forward_samples)
    # + one batched metric op; the two unfolded figures keep the reference's loop order (T walks, one model(X) per batch)
    for name, defer, fold in (("folded", True, True), ("deferred_sync", True, False), ("per_batch_sync", False, False)):
        loss = MultiExitAccuracy(n_exits)
        loss.defer_host_sync = defer
        evaluate(loss, loader, model, 0, "loop_bench", 1 if not fold else a.evaluate, create_log=False, fold=fold)          # warm-up
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            vec = evaluate(loss, loader, model, 0, "loop_bench", a.evaluate, create_log=False, fold=fold)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        out[name] = {"s": round(t, 4), "mcd_samples_per_s": round(a.images * a.evaluate / t, 1), "ms_per_model_call": round(1e3 * t / (a.evaluate * len(loader)), 4),
                     "acc1_avg": round(float(vec[0]), 6)}
    print(json.dumps({"what": "train.evaluate (SA/train/evaluate.py:8-22): T outer passes over a host loader, model(X) = one stochastic pass per call",
                      "workload": wl[5], "images": a.images, "batch": a.batch, "passes": a.evaluate, "pinned_host_batches": bool(a.pin), **out}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="resnet18_me", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--images", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=250)
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--pin", type=int, default=0,
                    help="pinned host batches (DataLoader pin_memory).  The reference's test loader pins (SA/datasets/dataset_loader.py:171); with 0 "
                         "workers the pinning copy runs on the loader's — this — thread and costs ~10 ms per 250-image batch on the GPU box's host "
                         "(tools/experiments/evaluate_profile.py): 0 is the loop at its best, 1 the reference's setting")
    ap.add_argument("--evaluate", type=int, default=0,
                    help="N > 0: time the reference's MCD use #1 instead — train.evaluate (SA/train/evaluate.py:8-22): N OUTER passes over the "
                         "loader, every model(X) ONE stochastic pass, the multi-exit accuracy vector per batch")
    ap.add_argument("--macro", type=int, default=1, help="FullAnalysis(macro_batches=K): K loader batches per engine step")
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
    T = a.T or wl[4]
    x = synthetic_images(a.images, seed=1234)                                   # host
    y = synthetic_labels(a.images, wl[2]["out_dim"], seed=1235)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=a.batch, shuffle=False, num_workers=0,
                                         pin_memory=bool(a.pin))
    if a.evaluate > 0:
        return evaluate_route(a, wl, model, loader)
    fa = FullAnalysis(model, None, gpu=0, mc_dropout=True, mc_passes=T, seed=42, macro_batches=a.macro)
    fa.loader = loader
    fa.sdn_get_detailed_results()                                                # warm-up: engines built, kernels loaded
    torch.cuda.synchronize()
    loop = []
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        fa.sdn_get_detailed_results()
        torch.cuda.synchronize()
        loop.append(time.perf_counter() - t0)
    # device-only: the same batches, resident in HBM, two in flight, results left on the device
    xd = x.to(dev)
    pipe = BatchesInFlight(model, dev, n=2, max_batch=a.batch)
    batches = [xd[i:i + a.batch] for i in range(0, a.images, a.batch)]
    for b in batches[:2]:
        pipe.submit(lambda e, b=b: e.predict(b, T, seed=42))
    pipe.synchronize()
    devonly = []
    for _ in range(a.repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, b in enumerate(batches):
            pipe.submit(lambda e, b=b, k=k: e.predict(b, T, seed=42 + k))
        pipe.synchronize()
        devonly.append(time.perf_counter() - t0)
    n = a.images * T
    tl, td = sorted(loop)[len(loop) // 2], sorted(devonly)[len(devonly) // 2]
    print(json.dumps({
        "what": "FullAnalysis loader walk (host loader -> H2D -> T passes -> host collation) vs the same batches device-resident",
        "workload": wl[5], "images": a.images, "batch": a.batch, "T": T, "pinned_host_batches": bool(a.pin), "macro_batches": a.macro,
        "pipe": {"in_flight": len(fa._pipe.engines), "hipgraph": bool(fa._pipe.use_graph), "engine_dtype": fa._pipe.engines[0].dtype},
        "loop_s": round(tl, 4), "loop_mcd_samples_per_s": round(n / tl, 1),
        "device_only_s": round(td, 4), "device_only_mcd_samples_per_s": round(n / td, 1),
        "loop_overhead_pct": round(100.0 * (tl - td) / td, 2),
        "all_loop_s": [round(v, 4) for v in loop], "all_device_only_s": [round(v, 4) for v in devonly],
        "accuracy_final_exit": float((fa.preds[-1].argmax(1) == fa.labels.argmax(1)).mean()),
    }), flush=True)


if __name__ == "__main__":
    main()
