#!/bin/bash
mkdir -p gpurun_out/r3_final
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_final/test_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r3_final/test_gpu.log
timeout 900 python tools/step_ab.py --workload resnet18_me --rounds 5 --steps 3 --ab "mask_lazy=1,mask_lazy=0" 2>&1 | tail -2 | tee gpurun_out/r3_final/ab18.log
timeout 900 python tools/step_ab.py --workload resnet50_me --rounds 5 --steps 3 --ab "mask_lazy=1,mask_lazy=0" 2>&1 | tail -2 | tee gpurun_out/r3_final/ab50.log
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"whole_step": {[^}]*}' | head -3
