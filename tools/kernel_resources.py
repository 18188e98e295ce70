#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
    python tools/kernel_resources.py conv3x3_patch.hip [-DNAME=V ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesnn_fpga_amd import _build  # noqa: E402

src = sys.argv[1]
path = src if os.path.exists(src) else os.path.join(_build.CSRC, src)
r = subprocess.run([_build._hipcc(), *_build.FLAGS, *sys.argv[2:], "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull],
                   capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
pat = (r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
       r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)")
print(f"{'kernel':90s} sgpr vgpr agpr scratch occ s-spill v-spill  lds")
for m in re.findall(pat, r.stderr, flags=re.S):
    name = subprocess.run(["c++filt", m[0]], capture_output=True, text=True).stdout.strip().replace("(ConvArgs)", "")
    print(f"{name[:90]:90s} {m[1]:>4s} {m[2]:>4s} {m[3]:>4s} {m[4]:>7s} {m[5]:>3s} {m[6]:>7s} {m[7]:>7s} {m[8]:>6s}")
