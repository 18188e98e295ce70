#!/bin/bash
# usage: tools/ab_run.sh "<command>" variant1 variant2 ...   (runs the command once per variant, twice over)
set -e
cmd="$1"; shift
cp bayesnn_fpga_amd/libbayesnn_fpga_amd.so /tmp/lib_orig.so
for rep in 1 2; do for v in "$@"; do cp bayesnn_fpga_amd/csrc/build/variants/lib_$v.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so; echo "== $v (rep $rep)"; eval "$cmd" 2>&1 | grep -v amdgpu.ids; done; done
cp /tmp/lib_orig.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
