#!/usr/bin/env python3
"""Builds several variants of the library (extra -D flags) into gpurun_out/variants/ for same-box A/B runs.
usage: python tools/ab_build.py name1:-DFOO=1,-DBAR=0 name2: ..."""
import os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesnn_fpga_amd import _build
out = os.path.join(_build.CSRC, "build", "variants")   # in-tree: gpurun ships it (gpurun_out/ is NOT shipped)
os.makedirs(out, exist_ok=True)
base = list(_build.FLAGS)
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    _build.FLAGS[:] = base + [f for f in flags.split(",") if f]
    _build.build(force=True)
    shutil.copy(_build.LIB, os.path.join(out, f"lib_{name}.so"))
    print("built", name, flags)
_build.FLAGS[:] = base
_build.build(force=True)
