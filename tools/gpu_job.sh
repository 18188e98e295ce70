#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_pool3
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/test_gpu.log 2>&1; echo "gpu tests rc=$?" >> $O/test_gpu.log
for i in 1 2 3; do timeout 600 python3 -m pytest tests/test_dynamic_exit.py tests/test_conv3x3_s2.py tests/test_race_screen.py -q >> $O/test_repeat.log 2>&1; done
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1
tail -5 $O/test_gpu.log; grep "passed\|failed" $O/test_repeat.log
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"whole_step": {[^}]*}' $O/bench.log
