#!/bin/bash
mkdir -p gpurun_out/r3_final
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_final/test_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3_final/test_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r3_final/bench.log 2>&1; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"whole_step": {[^}]*}' gpurun_out/r3_final/bench.log | head -3
