#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_direct
bash tools/ab_any.sh "python -m pytest tests/test_conv3x3_s2.py -x -q 2>&1 | tail -1; python tools/conv_bench.py --images 25000 --iters 10 --only D2p,D3,D3p,D4,D4p --nores --sparse-input 2>&1 | grep -v amdgpu | grep -v all" base direct > gpurun_out/r3_direct/ab.log 2>&1
cat gpurun_out/r3_direct/ab.log
