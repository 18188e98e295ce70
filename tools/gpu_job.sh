#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
cp profiles/hbm_traffic_resnet18_masksembles.json /tmp/keep.json 2>/dev/null
for i in 1 2 3; do python3 bench.py --workload resnet18_masksembles --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r03b/m_$i.json; python3 -c "
import json; d=json.load(open('gpurun_out/r03b/m_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['whole_step'])"; done
