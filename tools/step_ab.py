#!/usr/bin/env python3
"""Same-process, interleaved A/B of whole bench steps (one batch x T samples through the engine) under different
bmi_set_option sets — the comparison the CDNA guide asks for (rule 24: separate invocations add cross-process variance).

    python tools/step_ab.py --workload resnet18_me --rounds 7 --steps 3 \
        --ab "mfma_shape_patch=32+mfma_shape_wide=32,mfma_shape_patch=16+mfma_shape_wide=16"
Prints per variant the median / min ms per step and the per-family conv time of one profiled step.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bayesnn_fpga_amd import _lib  # noqa: E402
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_  # noqa: E402


ENGINES = []      # the live engines of this run: an option set is applied to the process defaults AND to their snapshots (C-ABI 600:
                  # bmi_create copies the defaults into the handle; bmi_engine_set_option edits one engine's copy)


def select(v):
    for kv in (v.split("+") if v else []):
        nm, _, val = kv.partition("=")
        _lib.set_option(nm, int(val))
        for e in ENGINES:
            e.set_option(nm, int(val))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="resnet18_me", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--ab", required=True)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
    B, T = a.batch or wl[3], a.T or wl[4]
    eng = model.engine(dev, max_batch=B, dtype="f16")
    ENGINES.append(eng)
    x = synthetic_images(B, seed=1234).to(dev)
    S = eng.new_moments(B)

    def step():
        S.zero_()
        eng.accumulate(x, S, 0, T, 42)
        return eng.finalize(S, T)

    variants = [v for v in a.ab.split(",")]
    times = {v: [] for v in variants}
    ref = None
    for v in variants:
        select(v)
        out = step()
        step()
        torch.cuda.synchronize()
        m = out["mean"].clone()
        if ref is None:
            ref = m
        print(f"{v}: max|mean - first variant's| = {float((m - ref).abs().max()):.2e}")
    for _ in range(a.rounds):
        for v in variants:
            select(v)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    for v in variants:
        select(v)
        eng.profile(True)
        step()
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        t = sorted(times[v])
        fam = "  ".join(f"{k.replace('_kernel', '')} {d['ms']:.2f} ms {d['flops'] / d['ms'] / 1e9:.0f} TF/s" for k, d in eng.conv_families.items())
        print(f"{v:60s} median {t[len(t) // 2]:8.3f} ms/step  min {t[0]:8.3f}   {B * T / t[len(t) // 2] * 1e3:10.0f} samples/s   | {fam} | "
              + " ".join(f"{k} {v2[0]:.2f}" for k, v2 in prof.items() if k != "conv_igemm"))


if __name__ == "__main__":
    main()
