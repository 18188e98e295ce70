#!/bin/bash
# usage (on the GPU box, from the repo root): [BENCH_EXTRA="--dtype f16x2"] tools/profile_bench.sh <workload> <tag> [<name of the hbm_traffic json, default = workload>]
# Kernel-trace stats pass + the two PMC passes of `python3 bench.py --workload W` -> gpurun_out/<tag>_*; then
# copy what is to be judged into profiles/ (see DESIGN.md §Measurement).
set -e
W=${1:-resnet18_me}; TAG=${2:-prof}; NAME=${3:-$W}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
# one batch at a time under the profiler: with two batches in flight (bench.py's default) the kernels of neighbouring batches share
# the GPU, and a kernel's trace duration / PMC window would include its neighbour's work
# --dtype f16 unless the caller names one: bench.py's default "auto" calibrates first (8 samples on two engines: small launches of the SAME kernels,
# which would drag the per-kernel averages of the trace down) and then runs exactly this engine on the synthetic bench weights
case "$BENCH_EXTRA" in *--dtype*) DT="" ;; *) DT="--dtype f16" ;; esac
ARGS="--workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-rccl-probe --no-parity-leg --no-graph --in-flight 1 $DT $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o run --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o run --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o run --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_sq -o run --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_sq.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/hbm_traffic_$NAME.json --note "bench.py $ARGS" --sq gpurun_out/${TAG}_sq
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
tail -1 gpurun_out/${TAG}_stats.log
