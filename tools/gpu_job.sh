#!/bin/bash
mkdir -p gpurun_out/r3_pool
cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py --workload resnet18_me --rounds 9 --steps 3 --ab "conv_pool=1,conv_pool=2" 2>&1 | tail -2 | tee gpurun_out/r3_pool/ab2.log
