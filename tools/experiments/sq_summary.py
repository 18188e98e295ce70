#!/usr/bin/env python3
"""usage: sq_summary.py <rocprofv3 --pmc SQ pass dir> : per kernel effective clock and MFMA-busy share (tools/pmc_summary.py's arithmetic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pmc_summary
for k, c in sorted(pmc_summary.fold_sq(sys.argv[1]).items()):
    if c["_ns"] <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0 or "conv" not in k:
        continue
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    wave = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print(f"{k[:70]:70s} launches {int(c['_launches']):4d}  clock {cycles / c['_ns']:.3f} GHz  mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024.0 * cycles):.3f}  "
          f"waiting {c.get('SQ_WAIT_ANY', 0) / wave:.3f} issue_stalled {c.get('SQ_WAIT_INST_ANY', 0) / wave:.3f} issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / wave:.3f}")
