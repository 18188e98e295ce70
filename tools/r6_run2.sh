#!/bin/bash
# round-6 GPU run 2: re-run of the tests that failed in run 1, the headline-size parity table (printed), ablation bounds for the
# fused-BasicBlock question, first lines of the three new workloads
mkdir -p gpurun_out/r6
python -m pytest tests/test_auto_engine.py -m gpu -q -s -p no:cacheprovider > gpurun_out/r6/parity_auto_engine.log 2>&1; echo "auto_engine rc=$?"
python -m pytest tests/test_converter.py tests/test_exact_engine.py tests/test_multi_gpu_mirrors.py tests/test_split_engine.py tests/test_host_api.py tests/test_collation.py -m gpu -q -p no:cacheprovider -rf > gpurun_out/r6/retest.log 2>&1; echo "retest rc=$?"; tail -3 gpurun_out/r6/retest.log
for W in resnet18_exit_only resnet18_layer vgg19_me; do
  python bench.py --workload $W --no-cpu-baseline --no-rccl-probe --no-parity-leg 2> gpurun_out/r6/${W}_bench.err | grep '^{' > gpurun_out/r6/${W}_bench_line.json; echo "$W rc=$?"
  python tools/per_launch.py --workload $W > gpurun_out/r6/${W}_per_launch.log 2>&1
done
python bench.py --workload resnet18_exit_only --macro 4 --no-cpu-baseline --no-rccl-probe --no-parity-leg 2>/dev/null | grep '^{' > gpurun_out/r6/resnet18_exit_only_macro4_bench_line.json
# ablation: what the 16x16-class conv would gain if its input came from cache / its output went nowhere (upper bound of a conv1->conv2 fusion in LDS)
{
for MOD in 25000 250 1; do
  echo "== plain conv (no residual), input tensor of $MOD images"; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --nores --in-mod $MOD
  echo "== tail conv (residual + 2-bit site), input tensor of $MOD images"; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --site --in-mod $MOD
done
tools/ab_any.sh "python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --nores; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --site; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --nores --in-mod 1; python tools/conv_bench.py --only S2 --images 25000 --iters 10 --rounds 3 --site --in-mod 1" base nostore nores
} > gpurun_out/r6/block_fusion_ablation.log 2>&1
echo done
