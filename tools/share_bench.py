"""One rank's share of a step on one GPU: how long rank 0 of `--world` ranks works per batch before the all-reduce, by partition
(samples | images) and launch mode (eager | one hipGraph replay), two batches in flight — what `sharding.partition` and
`bench.py --graph` decide between for the launch-bound shares (config 4: T = M = 8 masks over 8 GPUs gives each rank ONE sample
of 250 images, or all 8 samples of 31 images).

    python tools/share_bench.py --workload resnet18_masksembles --world 8 [--steps 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from bayesnn_fpga_amd.engine import BatchesInFlight  # noqa: E402
from bayesnn_fpga_amd.sharding import accumulate_share, partition  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="resnet18_masksembles", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--macro", type=int, default=1, help="loader batches carried by one step (a macro-batch of macro x 250 images): config 4's "
                                                        "share of eight is 31 images x 8 masks per 250-image batch — too little work per rank")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[a.workload]
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
    B, T = wl[3] * a.macro, a.T or wl[4]
    x = synthetic_images(B, seed=1234).to(dev)
    res = {}
    cases = [(kind, a.world) for kind in ("samples", "images")] + [("samples", 1)]      # (.., 1): the whole step on one GPU, the ratio's numerator
    for kind, world in cases:
        k, lo, hi = partition(T, B, 0, world, kind)
        if hi <= lo:
            continue
        for mode in ("eager", "graph"):
            pipe = BatchesInFlight(model, dev, n=a.in_flight, max_batch=B)
            Ss = [e.new_moments(B) for e in pipe.engines]

            def eager_step():
                i = pipe.slot()

                def run(e):
                    Ss[i].zero_()
                    accumulate_share(e, x, Ss[i], T, 42, 0, 0, world, kind)
                    return e.finalize(Ss[i], T)
                return pipe.submit(run)

            graphs = {}

            def graph_step():
                i = pipe.slot()
                pipe.k += 1
                e = pipe.engines[i]
                st = pipe.streams[i] or torch.cuda.current_stream(dev)
                if i not in graphs:
                    with torch.cuda.stream(st):
                        accumulate_share(e, x, Ss[i], T, 42, 0, 0, world, kind)
                    st.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        Ss[i].zero_()
                        accumulate_share(e, x, Ss[i], T, 42, 0, 0, world, kind)
                    graphs[i] = g
                with torch.cuda.stream(st):
                    graphs[i].replay()
                    return e.finalize(Ss[i], T)          # (the all-reduce would sit here)

            step = eager_step if mode == "eager" else graph_step
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            res[f"{kind}_{mode}" if world > 1 else f"whole_{mode}"] = round(ms, 4)
            print(f"{a.workload} world={world} rank 0: {kind} [{lo},{hi}) of T={T} B={B}  {mode:5}  {ms:.4f} ms per step", flush=True)
            del pipe, Ss, graphs
            torch.cuda.empty_cache()
    best = min(v for k, v in res.items() if not k.startswith("whole"))
    whole = min(v for k, v in res.items() if k.startswith("whole"))
    print(json.dumps({"workload": a.workload, "world": a.world, "T": T, "batch": B, "macro_batches": a.macro, "in_flight": a.in_flight, "steps": a.steps,
                      "ms_per_step": res, "implied_ratio_before_allreduce": round(whole / best, 2),
                      "note": "whole_* = the same macro-batch on ONE rank; ratio = whole / the fastest share of rank 0 (no collective in either)"}))


if __name__ == "__main__":
    main()
