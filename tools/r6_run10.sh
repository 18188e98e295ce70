#!/bin/bash
# round-6 GPU run 10: lazy keep-bit site for the stride-1 readers of layer mode (64-channel tile masks its patch and its residual)
mkdir -p gpurun_out/r6
python -m pytest tests/test_full_batch.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_dynamic_exit.py tests/test_race_screen.py -m gpu -q --maxfail=30 -rf -p no:cacheprovider -s > gpurun_out/r6/gpu_tests_10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_10.log; grep "first MASK" gpurun_out/r6/gpu_tests_10.log
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
for rep in 1 2; do for L in 1 0; do BMI_OPTIONS="mask_lazy=$L" python bench.py $B --workload resnet18_layer 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('layer mask_lazy=$L', d['value'], d['ms_per_step'], d['roofline']['whole_step']['frac'])"; done; done
python tools/per_launch.py --workload resnet18_layer 2>/dev/null | head -12
python bench.py $B --workload resnet18_me 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['whole_step']['frac'])"
echo done
