#!/bin/bash
# round-6 GPU run 7: where the exit-only loader walk spends its host time (cProfile), pair_prefix default across the workloads
mkdir -p gpurun_out/r6
python -m cProfile -s cumtime tools/loop_bench.py --workload resnet18_exit_only --macro 4 --repeats 2 2>/dev/null | head -70 > gpurun_out/r6/cprofile_loop_macro4.txt
python -m cProfile -s cumtime tools/loop_bench.py --workload resnet18_exit_only --macro 1 --repeats 2 2>/dev/null | head -70 > gpurun_out/r6/cprofile_loop_macro1.txt
python tools/loop_bench.py --workload vgg19_me 2>&1 | tail -5 > gpurun_out/r6/loop2_vgg19_me.txt
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
run() { tag=$1; opts=$2; shift; shift; BMI_OPTIONS="$opts" python bench.py $B "$@" 2>/dev/null | grep '^{' > gpurun_out/r6/${tag}.json; python - <<PY
import json
d=json.load(open("gpurun_out/r6/${tag}.json")); print("${tag}", d["value"], d["ms_per_step"], d["config"]["pipe"][:12], d["config"]["rank_step_probe_ms"], d["roofline"]["whole_step"]["frac"])
PY
}
for W in vgg19_me vgg11 resnet18_masksembles resnet50_me resnet18_me; do
run z_${W}_pp0 "pair_prefix=0" --workload $W
run z_${W}_pp1 "pair_prefix=1" --workload $W
done
python -m pytest tests/test_full_batch.py tests/test_gpu_model.py tests/test_vgg.py tests/test_extra_models.py tests/test_split_engine.py tests/test_exact_engine.py tests/test_converter.py tests/test_collation.py -m gpu -q --maxfail=30 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_7.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_7.log
echo done
