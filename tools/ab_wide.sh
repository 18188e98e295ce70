#!/bin/bash
# same-box A/B of conv_igemm_wide variants: tools/ab_wide.sh base pipe ...   (variants from tools/ab_build.py)
cp bayesnn_fpga_amd/libbayesnn_fpga_amd.so /tmp/lib_orig.so
cp /tmp/lib_orig.so bayesnn_fpga_amd/csrc/build/variants/lib_base.so
for rep in 1 2; do for v in "$@"; do cp bayesnn_fpga_amd/csrc/build/variants/lib_$v.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so; echo "== $v (rep $rep)"; timeout 300 python tools/conv_bench.py --images 25000 --iters 10 --only D3,D4,P4 --nores 2>&1 | grep -v amdgpu; done; done
cp /tmp/lib_orig.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
