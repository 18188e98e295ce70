#!/bin/bash
# round-6 GPU run 5: full GPU suite on the final code, default bench line, loader-walk (FullAnalysis) figures for the paper's configuration
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q --maxfail=40 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_5.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_5.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r6/bench_5.json 2> gpurun_out/r6/bench_5.err; echo "bench rc=$?"
for K in 1 4 8; do python tools/loop_bench.py --workload resnet18_exit_only --macro $K 2>/dev/null | grep '^{' > gpurun_out/r6/loop_exit_only_macro$K.json; done
python tools/loop_bench.py --workload resnet18_me --images 5000 2>/dev/null | grep '^{' > gpurun_out/r6/loop_resnet18_me.json
python tools/loop_bench.py --workload resnet18_exit_only --evaluate 10 2>/dev/null | grep '^{' > gpurun_out/r6/evaluate_exit_only.json
cat gpurun_out/r6/loop_*.json gpurun_out/r6/evaluate_exit_only.json | cut -c1-900
