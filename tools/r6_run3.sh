#!/bin/bash
# round-6 GPU run 3: the full GPU suite after the conv_split row tables + batched heads, and the pipe / macro variants of the exit-only lines
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q --maxfail=40 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_3.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_3.log
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
run() { tag=$1; shift; python bench.py $B "$@" 2>/dev/null | grep '^{' > gpurun_out/r6/${tag}.json; python - <<PY
import json
d=json.load(open("gpurun_out/r6/${tag}.json")); print("${tag}", d["value"], d["ms_per_step"], d["config"]["pipe"][:30], d["roofline"]["whole_step"]["frac"])
PY
}
for rep in 1 2; do
run exit_only_default_$rep --workload resnet18_exit_only
run exit_only_graph3_$rep --workload resnet18_exit_only --in-flight 3 --graph
run exit_only_eager3_$rep --workload resnet18_exit_only --in-flight 3 --no-graph
run exit_only_macro4_$rep --workload resnet18_exit_only --macro 4
run exit_only_macro4_graph3_$rep --workload resnet18_exit_only --macro 4 --in-flight 3 --graph
run exit_only_macro8_$rep --workload resnet18_exit_only --macro 8
run vgg19_default_$rep --workload vgg19_me
run vgg19_graph3_$rep --workload vgg19_me --in-flight 3 --graph
run masks_default_$rep --workload resnet18_masksembles
run masks_graph3_$rep --workload resnet18_masksembles --in-flight 3 --graph
done
python tools/per_launch.py --workload resnet18_exit_only > gpurun_out/r6/resnet18_exit_only_per_launch_batched.log 2>&1
echo done
