#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_misc1
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_collation.py -q -x > $O/test_model.log 2>&1; echo "rc=$?" >> $O/test_model.log
python3 tools/loop_bench.py > $O/loop_r18.log 2>&1
python3 tools/loop_bench.py --pin 0 > $O/loop_r18_nopin.log 2>&1
python3 bench.py --workload vgg11 --steps 200 --warmup 20 --no-cpu-baseline > $O/vgg11_eager.log 2>&1
python3 bench.py --workload vgg11 --steps 200 --warmup 20 --no-cpu-baseline --graph > $O/vgg11_graph.log 2>&1
python3 bench.py --workload vgg11 --steps 200 --warmup 20 --no-cpu-baseline --graph --in-flight 3 > $O/vgg11_graph3.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1024 --T 25 > $O/r18_b1024.log 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --graph > $O/r18_graph.log 2>&1
tail -3 $O/test_model.log; tail -1 $O/loop_r18.log; tail -1 $O/loop_r18_nopin.log
for f in vgg11_eager vgg11_graph vgg11_graph3 r18_b1024 r18_graph; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/$f.log | tr '\n' ' '; echo; done
