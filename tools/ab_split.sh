#!/bin/bash
# usage (build container): tools/ab_split.sh name1:-DFOO=1 name2:-DBAR=1,-DBAZ=2 ...
# Builds variants of the library that differ in csrc/conv_split.hip's -D flags only (the other objects are reused) into
# bayesnn_fpga_amd/csrc/build/variants/lib_<name>.so, for same-box A/B runs with tools/ab_run.sh.
set -e
cd "$(dirname "$0")/.."
python -m bayesnn_fpga_amd._build > /dev/null
B=bayesnn_fpga_amd/csrc/build
mkdir -p $B/variants
OTHERS=$(ls $B/*.o | grep -v conv_split.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $flags -c bayesnn_fpga_amd/csrc/conv_split.hip -o $B/variants/conv_split_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/lib_$name.so $OTHERS $B/variants/conv_split_$name.o
  echo "built $name ($flags)"
done
