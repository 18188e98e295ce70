#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_wsplit
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_converter.py tests/test_gpu_model.py -q -m gpu > $O/test_new.log 2>&1; echo "new tests rc=$?" >> $O/test_new.log
tools/ab_any.sh "python3 tools/conv_bench.py --images 12000 --iters 10 --rounds 3 --nores --sparse-input --only S3,S4,D2p,D3,D3p,D4,D4p" base wsplit > $O/ab.log 2>&1
cp bayesnn_fpga_amd/libbayesnn_fpga_amd.so /tmp/lib_base.so
cp bayesnn_fpga_amd/csrc/build/variants/lib_wsplit.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
timeout 1500 python3 -m pytest tests/test_conv3x3_s2.py tests/test_race_screen.py tests/test_gpu_kernels.py -q > $O/test_wsplit.log 2>&1; echo "wsplit tests rc=$?" >> $O/test_wsplit.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_wsplit.log 2>&1
cp /tmp/lib_base.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_base.log 2>&1
tail -4 $O/test_new.log; tail -3 $O/test_wsplit.log; grep "==\|median" $O/ab.log | cut -c1-100
for f in bench_base bench_wsplit; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/$f.log | tr '\n' ' '; echo; done
