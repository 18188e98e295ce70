#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_s2c
mkdir -p $O
timeout 900 python3 -m pytest tests/test_conv3x3_s2.py tests/test_race_screen.py -q > $O/test_s2.log 2>&1; echo "s2+race tests rc=$?" >> $O/test_s2.log
python3 tools/per_launch.py --workload resnet18_me > $O/per_launch_r18.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench2.log 2>&1
tail -3 $O/test_s2.log; grep "conv3x3_s2" $O/per_launch_r18.log; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"whole_step": {[^}]*}' $O/bench.log $O/bench2.log
