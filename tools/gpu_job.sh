#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_fix
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q -x > $O/test_gpu.log 2>&1; echo "gpu tests rc=$?" >> $O/test_gpu.log
tail -6 $O/test_gpu.log
