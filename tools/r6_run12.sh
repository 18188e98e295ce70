#!/bin/bash
# round-6 GPU run 12: conv3x3_patch's register epilogue for the 16x16 class ("patch_direct"): bit-identity tests, same-process step A/B, per-launch table
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py tests/test_full_batch.py tests/test_gpu_model.py tests/test_race_screen.py tests/test_dynamic_exit.py -m gpu -q --maxfail=20 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_12.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r6/gpu_tests_12.log
python tools/step_ab.py --workload resnet18_me --rounds 7 --steps 3 --ab "patch_direct=0,patch_direct=1" 2>/dev/null | grep -v amdgpu > gpurun_out/r6/patch_direct_step_ab.log; cat gpurun_out/r6/patch_direct_step_ab.log | cut -c1-260
python tools/step_ab.py --workload resnet18_masksembles --rounds 7 --steps 5 --ab "patch_direct=0,patch_direct=1" 2>/dev/null | grep -v amdgpu >> gpurun_out/r6/patch_direct_step_ab.log; tail -2 gpurun_out/r6/patch_direct_step_ab.log | cut -c1-200
for A in 0 1; do echo "== patch_direct=$A"; python tools/per_launch.py --workload resnet18_me --set patch_direct=$A 2>/dev/null | sed -n 13,16p; done
{ for A in "--nores" "" "--site"; do echo "== S2 25000 images $A"; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 $A --ab "patch_direct=0,patch_direct=1"; done; } 2>&1 | grep -v amdgpu | tee gpurun_out/r6/patch_direct_conv_bench.log
echo done
