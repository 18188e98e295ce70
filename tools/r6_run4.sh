#!/bin/bash
# round-6 GPU run 4: the 64-channel patch tile (unit + engine tests), split-K threshold A/B, exit-only / layer / VGG-19 lines
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_full_batch.py tests/test_vgg.py tests/test_race_screen.py -m gpu -q --maxfail=30 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_4.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_4.log
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
run() { tag=$1; opts=$2; shift; shift; BMI_OPTIONS="$opts" python bench.py $B "$@" 2>/dev/null | grep '^{' > gpurun_out/r6/${tag}.json; python - <<PY
import json
d=json.load(open("gpurun_out/r6/${tag}.json")); print("${tag}", d["value"], d["ms_per_step"], d["config"]["pipe"][:12], d["config"]["rank_step_probe_ms"], d["roofline"]["whole_step"]["frac"])
PY
}
for rep in 1 2; do
run x_exit_p64_0_sk64_$rep "conv_patch64=0,splitk_tiles=64" --workload resnet18_exit_only
run x_exit_p64_1_sk64_$rep "conv_patch64=1,splitk_tiles=64" --workload resnet18_exit_only
run x_exit_p64_1_sk128_$rep "conv_patch64=1,splitk_tiles=128" --workload resnet18_exit_only
run x_exit_p64_1_sk256_$rep "conv_patch64=1,splitk_tiles=256" --workload resnet18_exit_only
run x_exit_macro8_p64_1_$rep "conv_patch64=1" --workload resnet18_exit_only --macro 8
run x_layer_p64_0_$rep "conv_patch64=0" --workload resnet18_layer
run x_layer_p64_1_$rep "conv_patch64=1" --workload resnet18_layer
run x_vgg19_p64_0_$rep "conv_patch64=0" --workload vgg19_me
run x_vgg19_p64_1_$rep "conv_patch64=1,splitk_tiles=64" --workload vgg19_me
run x_vgg19_p64_1_sk128_$rep "conv_patch64=1,splitk_tiles=128" --workload vgg19_me
run x_head_p64_0_$rep "conv_patch64=0" --workload resnet18_me
run x_head_p64_1_$rep "conv_patch64=1" --workload resnet18_me
done
python tools/per_launch.py --workload resnet18_exit_only > gpurun_out/r6/resnet18_exit_only_per_launch_p64.log 2>&1
BMI_OPTIONS="splitk_tiles=128" python tools/per_launch.py --workload resnet18_exit_only > gpurun_out/r6/resnet18_exit_only_per_launch_p64_sk128.log 2>&1
python tools/per_launch.py --workload resnet18_layer > gpurun_out/r6/resnet18_layer_per_launch_p64.log 2>&1
echo done
