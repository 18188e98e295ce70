#!/bin/bash
mkdir -p gpurun_out/r3_lazy
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_lazy/test_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/r3_lazy/test_gpu.log
