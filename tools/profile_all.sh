#!/bin/bash
# usage (GPU box, repo root): tools/profile_all.sh <tag> [workload ...]
# For every workload: bench line (with cpu baseline for the headline only), rocprofv3 kernel stats, the two HBM PMC passes and
# the SQ pass (tools/profile_bench.sh), collected under gpurun_out/<tag>/ ready to be copied into profiles/.
TAG=${1:-r02}; shift
WL=${@:-resnet18_me vgg11 resnet18_masksembles resnet50_me}
mkdir -p gpurun_out/$TAG
for W in $WL; do
  tools/profile_bench.sh $W ${TAG}_$W > gpurun_out/$TAG/${W}_profile.log 2>&1
  cp gpurun_out/hbm_traffic_$W.json gpurun_out/$TAG/hbm_traffic_$W.json
  cp gpurun_out/${TAG}_${W}_kernel_stats.csv gpurun_out/$TAG/${TAG}_${W}_kernel_stats.csv
  # the bench line quotes profiles/hbm_traffic_<W>.json when its launch count matches: refresh it first
  cp gpurun_out/hbm_traffic_$W.json profiles/hbm_traffic_$W.json
  EXTRA="--no-cpu-baseline --no-parity-leg --no-rccl-probe"; [ "$W" = resnet18_me ] && EXTRA=""
  python3 bench.py --workload $W --steps 10 --warmup 3 $EXTRA 2> /dev/null | grep '^{' > gpurun_out/$TAG/${TAG}_${W}_bench_line.json
  rm -rf gpurun_out/${TAG}_${W}_stats gpurun_out/${TAG}_${W}_fetch gpurun_out/${TAG}_${W}_write gpurun_out/${TAG}_${W}_sq
done
ls -la gpurun_out/$TAG
