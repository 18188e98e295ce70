#!/bin/bash
# usage (build container): tools/ab_one.sh <source without .hip> name1:-DFOO=1 name2:-DBAR=1,-DBAZ=2 ...
# Builds variants of the library that differ in csrc/$SRC.hip's -D flags only (the other objects are reused) into
# bayesnn_fpga_amd/csrc/build/variants/lib_<name>.so, for same-box A/B runs with tools/ab_run.sh.
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
python -m bayesnn_fpga_amd._build > /dev/null
B=bayesnn_fpga_amd/csrc/build
mkdir -p $B/variants
OTHERS=$(ls $B/*.o | grep -v $SRC.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $flags -c bayesnn_fpga_amd/csrc/$SRC.hip -o $B/variants/${SRC}_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $B/variants/lib_$name.so $OTHERS $B/variants/${SRC}_$name.o
  echo "built $name ($flags)"
done
