#!/usr/bin/env python3
"""Per-launch table of one bench step: every launch of the engine (in order) with its shape, kernel family, device time and
its rate against the two rooflines (algorithmic FLOPs / bytes from the engine, HIP-event times).

    python tools/per_launch.py --workload resnet50_me [--T 64] [--batch 250] [--top 15]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesnn_fpga_amd import _lib  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="resnet18_me", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--top", type=int, default=0, help="also list the N launches with the most time")
    ap.add_argument("--set", default="", help="bmi_set_option pairs, name=value+name=value")
    ap.add_argument("--dtype", default="f16", choices=sorted(_lib.DTYPES))
    a = ap.parse_args()
    for kv in (a.set.split("+") if a.set else []):
        nm, _, val = kv.partition("=")
        _lib.set_option(nm, int(val))
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
    B, T = a.batch or wl[3], a.T or wl[4]
    eng = model.engine(dev, max_batch=B, dtype=a.dtype)
    x = synthetic_images(B, seed=1234).to(dev)
    S = eng.new_moments(B)
    for _ in range(2):
        S.zero_()
        eng.accumulate(x, S, 0, T, 42)
    torch.cuda.synchronize()
    eng.profile(True)
    S.zero_()
    eng.accumulate(x, S, 0, T, 42)
    torch.cuda.synchronize()
    eng.profile_read()
    rows = eng.profile_launches()
    eng.profile(False)
    g = eng.graph
    by_out = {op["out"]: op for op in g.ops if op.get("out", -1) >= 0 and "ksize" in op}   # conv ops (heads / masks print no shape)
    total = sum(r["ms"] for r in rows)
    print(f"{a.workload}: batch {B} x T {T}, {len(rows)} launches, {total:.2f} ms of device time")
    print(f"{'#':>3} {'kind':7} {'family':16} {'shape':34} {'images':>7} {'ms':>7} {'TFLOP/s':>8} {'GB/s':>7}  flags")
    lines = []
    for i, r in enumerate(rows):
        op = by_out.get(r["out"]) if r["kind"] in ("conv_igemm", "conv", "stem") else None
        shape, flags = "", ""
        if op is not None and "in_" in op:
            h, w, c = g.tensors[op["in_"]]
            ho, wo, co = g.tensors[op["out"]]
            k, s = op.get("ksize", 0), op.get("stride", 0)
            shape = f"{c}->{co} k{k}s{s} {h}x{w}->{ho}x{wo}" if k else f"{c}@{h}x{w}"
            flags = " ".join(f for f, on in (("res", op.get("residual", -1) >= 0), ("site", op.get("site") is not None),
                                             ("shortcut", op.get("in2", -1) >= 0)) if on)
        tf = r["flops"] / r["ms"] / 1e9 if r["ms"] > 0 and r["flops"] else 0.0
        gb = r["bytes"] / r["ms"] / 1e6 if r["ms"] > 0 and r["bytes"] else 0.0
        kind = "conv" if r["kind"] == "conv_igemm" else r["kind"]      # (the library's slot name for every conv op)
        line = f"{i:3d} {kind:7} {r['family'] or '':16} {shape:34} {r['images']:7d} {r['ms']:7.3f} {tf:8.1f} {gb:7.0f}  {flags}"
        lines.append((r["ms"], line))
        print(line)
    if a.top:
        print(f"--- top {a.top} by time")
        for _, line in sorted(lines, reverse=True)[:a.top]:
            print(line)


if __name__ == "__main__":
    main()
