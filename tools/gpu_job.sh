#!/bin/bash
mkdir -p gpurun_out/r3_graph
cd $GRAFT_REPO_ROOT
for ARGS in "--in-flight 2" "--in-flight 3" "--graph --in-flight 2" "--graph --in-flight 3"; do
echo "masksembles $ARGS"; python3 bench.py --workload resnet18_masksembles --steps 100 --warmup 10 --no-cpu-baseline $ARGS 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/r3_graph/masksembles.log
python3 bench.py --workload resnet18_masksembles --steps 100 --warmup 10 --no-cpu-baseline --graph --in-flight 3 2>/dev/null | grep '^{' > gpurun_out/r3_graph/r03_resnet18_masksembles_graph_bench_line.json
