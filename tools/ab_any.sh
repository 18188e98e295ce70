#!/bin/bash
# same-box A/B of library variants built by tools/ab_build.py: tools/ab_any.sh "<command>" base variant...
cmd="$1"; shift
cp bayesnn_fpga_amd/libbayesnn_fpga_amd.so /tmp/lib_orig.so
cp /tmp/lib_orig.so bayesnn_fpga_amd/csrc/build/variants/lib_base.so
for rep in 1 2; do for v in "$@"; do cp bayesnn_fpga_amd/csrc/build/variants/lib_$v.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so; echo "== $v (rep $rep)"; eval "$cmd" 2>&1 | grep -v amdgpu; done; done
cp /tmp/lib_orig.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
