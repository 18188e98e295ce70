#!/bin/bash
# round-6 GPU run 9: the 64-channel tile's register-form epilogue (A/B against the previous build), layer mode, remaining profiles
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py tests/test_auto_engine.py -m gpu -q --maxfail=30 -rf -p no:cacheprovider -k "patch_kernel_64 or shape_classes or auto" > gpurun_out/r6/gpu_tests_9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6/gpu_tests_9.log
{
for A in "" "--nores" "--site" ; do echo "== S1 25000 images $A"; python tools/conv_bench.py --only S1 --images 25000 --iters 5 --rounds 3 $A --ab "conv_patch64=1,conv_patch64=0"; done
} > gpurun_out/r6/patch64_epilogue.log 2>&1; grep -v amdgpu gpurun_out/r6/patch64_epilogue.log
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
for rep in 1 2; do python bench.py $B --workload resnet18_layer 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('layer', d['value'], d['ms_per_step'], d['roofline']['whole_step']['frac'])"; done
python bench.py $B --workload vgg11 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vgg11 auto', d['value'], d['ms_per_step'], d['config']['engine_dtype'])"
tools/profile_all.sh r06 vgg11 resnet18_masksembles resnet50_me resnet18_layer resnet18_me
echo done
