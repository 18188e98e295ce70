#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3 + 2)
of `python3 bench.py ...` into profiles/<name>.json: HBM-side bytes per launch for every kernel.

Corrections, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

An optional third pass (<sq_dir>: SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) adds, per kernel: the effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / launch
duration), MFMA-pipe busy share (MFMA_BUSY cycles / (1024 SIMDs x launch cycles)) and the wave-cycle split
(waiting at s_waitcnt or barrier / issue-stalled / issuing), as MI355X_MICROARCH.md "rocprofv3 PMC slots" defines them.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [--note "..."] [--sq <sq_dir>]
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


def fold_sq(d):
    """per kernel: sums of every counter and of the launch durations (ns) over the launches"""
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "")
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    acc[k]["_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    acc[k]["_launches"] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    sq_dir = sys.argv[sys.argv.index("--sq") + 1] if "--sq" in sys.argv else None
    fe, wr = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        n = fe.get(k, wr.get(k))[0]
        fb = 2.0 * 1024.0 * fe[k][1] / max(fe[k][0], 1) if k in fe else None
        wb = 1024.0 * wr[k][1] / max(wr[k][0], 1) if k in wr else None
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0)}
    if sq_dir:
        for k, c in fold_sq(sq_dir).items():
            if k not in kernels or c["_ns"] <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
                continue
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
            wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            kernels[k]["sq"] = {
                "effective_clock_ghz": round(cycles / c["_ns"], 3),
                "mfma_busy_share": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cycles), 4),
                "wave_cycles_waiting": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4),
                "wave_cycles_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
                "wave_cycles_issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4),
            }
    json.dump({"note": note, "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950); WRITE_SIZE as read",
               "kernels": kernels}, open(out, "w"), indent=1)
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kernels.values())
    print(f"{len(kernels)} kernels, {tot / 1e9:.3f} GB over all launches -> {out}")


if __name__ == "__main__":
    main()
