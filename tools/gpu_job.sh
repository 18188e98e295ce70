#!/bin/bash
mkdir -p gpurun_out/r3_lds
cd $GRAFT_REPO_ROOT
bash tools/ab_any.sh "python tools/conv_bench.py --images 8000 --iters 10 --only S3,S4,D3p,D4p --nores --sparse-input 2>&1 | grep -v amdgpu | grep -v all" base half > gpurun_out/r3_lds/half.log 2>&1
cat gpurun_out/r3_lds/half.log
