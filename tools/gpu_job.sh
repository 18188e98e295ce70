#!/bin/bash
mkdir -p gpurun_out/r3_pw4
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pw4" > gpurun_out/r3_pw4/t.log 2>&1; tail -3 gpurun_out/r3_pw4/t.log
