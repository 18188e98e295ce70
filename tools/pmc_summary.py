#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3 + 2)
of `python3 bench.py ...` into profiles/<name>.json: HBM-side bytes per launch for every kernel.

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [--note "..."]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def fold(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--note" else ""
    fe, wr = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        n = fe.get(k, wr.get(k))[0]
        fb = 2.0 * 1024.0 * fe[k][1] / max(fe[k][0], 1) if k in fe else None
        wb = 1024.0 * wr[k][1] / max(wr[k][0], 1) if k in wr else None
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0)}
    json.dump({"note": note, "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950); WRITE_SIZE as read",
               "kernels": kernels}, open(out, "w"), indent=1)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kernels.values())
    print(f"{len(kernels)} kernels, {tot / 1e9:.3f} GB over all launches -> {out}")


if __name__ == "__main__":
    main()
