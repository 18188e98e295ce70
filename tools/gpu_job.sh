#!/bin/bash
# final round-3 profiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03 gpurun_out/r3_prof5
bash tools/profile_all.sh r03 > gpurun_out/r3_prof5/profile_all.log 2>&1
python3 bench.py --workload vgg11 --steps 200 --warmup 20 --no-cpu-baseline --graph --in-flight 3 2>/dev/null | grep '^{' > gpurun_out/r03/r03_vgg11_graph_bench_line.json
python3 bench.py --workload vgg11 --steps 200 --warmup 20 --no-cpu-baseline --in-flight 3 2>/dev/null | grep '^{' > gpurun_out/r03/r03_vgg11_inflight3_bench_line.json
python3 bench.py --batch 1024 --T 25 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03/r03_resnet18_me_b1024_bench_line.json
python3 tools/loop_bench.py --pin 0 2>/dev/null | grep '^{' > gpurun_out/r03/r03_loop_bench_line.json
python3 tools/per_launch.py --workload resnet18_me > gpurun_out/r03/r03_resnet18_me_per_launch.log 2>&1
python3 tools/per_launch.py --workload resnet50_me > gpurun_out/r03/r03_resnet50_me_per_launch.log 2>&1
for T in 13 25 50 100; do echo -n "T=$T "; python3 bench.py --T $T --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done > gpurun_out/r03/r03_t_share_ms.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_prof5/smoke.log 2>&1; tail -1 gpurun_out/r3_prof5/smoke.log
for f in gpurun_out/r03/*bench_line.json; do echo $f; python3 -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{})
print(d.get('value'), d.get('ms_per_step'), r.get('whole_step'), r.get('kernel'), r.get('frac'), d.get('loop_mcd_samples_per_s'), d.get('device_only_mcd_samples_per_s'))"; done
cat gpurun_out/r03/r03_t_share_ms.txt
