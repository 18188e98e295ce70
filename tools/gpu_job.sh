#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_prof3
mkdir -p $O
rm -rf gpurun_out/r03
tools/profile_all.sh r03 > $O/profile_all.log 2>&1
python3 bench.py --workload vgg11 --steps 300 --warmup 30 --no-cpu-baseline --graph --in-flight 3 2> /dev/null | grep '^{' > gpurun_out/r03/r03_vgg11_graph_bench_line.json
python3 bench.py --workload vgg11 --steps 300 --warmup 30 --no-cpu-baseline --in-flight 3 2> /dev/null | grep '^{' > gpurun_out/r03/r03_vgg11_inflight3_bench_line.json
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1024 --T 25 2> /dev/null | grep '^{' > gpurun_out/r03/r03_resnet18_me_b1024_bench_line.json
python3 tools/loop_bench.py --pin 0 2> /dev/null | grep '^{' > gpurun_out/r03/r03_loop_bench_line.json
python3 tools/per_launch.py --workload resnet18_me > gpurun_out/r03/r03_resnet18_me_per_launch.log 2>&1
python3 tools/per_launch.py --workload resnet50_me > gpurun_out/r03/r03_resnet50_me_per_launch.log 2>&1
for T in 13 25 50 100; do python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --T $T 2> /dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/T=$T /"; done > gpurun_out/r03/r03_t_share_ms.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for f in gpurun_out/r03/*bench_line.json; do echo $f; python3 -c "
import json,sys
d=json.load(open('$f'))
cb=d.get('cpu_baseline') or {}
print(d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('whole_step'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), cb.get('max_abs_mean_diff_gpu_vs_cpu'), cb.get('value'))
"; done; cat gpurun_out/r03/r03_t_share_ms.txt
