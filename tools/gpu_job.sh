#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_tj1
BMI_STREAM_TJ1=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv1x1_stream" 2>&1 | tail -2
for i in 1 2; do
python tools/conv_bench.py --images 16000 --iters 10 --only Q1,R2,R3,E2,E3 --nores --sparse-input --ab conv_stream=2 2>&1 | grep -v amdgpu | grep -v all
BMI_STREAM_TJ1=1 python tools/conv_bench.py --images 16000 --iters 10 --only Q1,R2,R3,E2,E3 --nores --sparse-input --ab conv_stream=2 2>&1 | grep -v amdgpu | grep -v all | sed 's/^/TJ1 /'
done | tee gpurun_out/r3_tj1/bench.log
