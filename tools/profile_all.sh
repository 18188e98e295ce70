#!/bin/bash
# usage (GPU box, repo root): tools/profile_all.sh <tag> [workload ...]
# For every workload: rocprofv3 kernel stats, the two HBM PMC passes and the SQ pass (tools/profile_bench.sh), the per-launch table
# (tools/per_launch.py) and the un-profiled bench line (with cpu baseline / parity leg / RCCL probe for the headline only), collected under
# gpurun_out/<tag>/ ready to be copied into profiles/.  vgg11 is profiled on the fp16 engine (comparable with the earlier rounds); what the
# product's default — engine_dtype="auto" — picks and measures on it is a second bench line (<tag>_vgg11_auto_bench_line.json).
TAG=${1:-r02}; shift
WL=${@:-resnet18_me resnet18_exit_only resnet18_layer vgg19_me vgg11 resnet18_masksembles resnet50_me}
mkdir -p gpurun_out/$TAG
for W in $WL; do
  DT=""; [ "$W" = vgg11 ] && DT="--dtype f16"
  BENCH_EXTRA="$DT" tools/profile_bench.sh $W ${TAG}_$W > gpurun_out/$TAG/${W}_profile.log 2>&1
  cp gpurun_out/hbm_traffic_$W.json gpurun_out/$TAG/hbm_traffic_$W.json
  cp gpurun_out/${TAG}_${W}_kernel_stats.csv gpurun_out/$TAG/${TAG}_${W}_kernel_stats.csv
  # the bench line quotes profiles/hbm_traffic_<W>.json and the newest profiles/rNN_<W>_kernel_stats.csv when their launch counts match: refresh them first
  cp gpurun_out/hbm_traffic_$W.json profiles/hbm_traffic_$W.json
  cp gpurun_out/${TAG}_${W}_kernel_stats.csv profiles/${TAG}_${W}_kernel_stats.csv
  EXTRA="--no-cpu-baseline --no-parity-leg --no-rccl-probe"; [ "$W" = resnet18_me ] && EXTRA=""
  python3 bench.py --workload $W --steps 10 --warmup 3 $EXTRA $DT 2> /dev/null | grep '^{' > gpurun_out/$TAG/${TAG}_${W}_bench_line.json
  [ "$W" = vgg11 ] && python3 bench.py --workload $W --steps 10 --warmup 3 $EXTRA 2> /dev/null | grep '^{' > gpurun_out/$TAG/${TAG}_${W}_auto_bench_line.json
  python3 tools/per_launch.py --workload $W 2>/dev/null > gpurun_out/$TAG/${TAG}_${W}_per_launch.log
  rm -rf gpurun_out/${TAG}_${W}_stats gpurun_out/${TAG}_${W}_fetch gpurun_out/${TAG}_${W}_write gpurun_out/${TAG}_${W}_sq
done
ls -la gpurun_out/$TAG
