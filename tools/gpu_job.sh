#!/bin/bash
mkdir -p gpurun_out/r3_shape
cd $GRAFT_REPO_ROOT
timeout 900 python tools/conv_bench.py --images 25000 --iters 10 --only S2 --nores --sparse-input --rounds 3 --ab mfma_shape_patch=16,mfma_shape_patch=32 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r3_shape/s2.log
