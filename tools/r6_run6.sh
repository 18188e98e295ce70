#!/bin/bash
# round-6 GPU run 6: loader-walk figures with the pipe cached on the model, prefix pair fusion / conv_s2 A/B on the exit-only line, affected tests
mkdir -p gpurun_out/r6
python -m pytest tests/test_collation.py tests/test_multi_gpu_mirrors.py tests/test_auto_engine.py tests/test_host_api.py tests/test_converter.py -m gpu -q --maxfail=30 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_6.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_6.log
for K in 1 4 8; do python tools/loop_bench.py --workload resnet18_exit_only --macro $K 2>/dev/null | grep '^{' > gpurun_out/r6/loop2_exit_only_macro$K.json; done
python tools/loop_bench.py --workload resnet18_me --images 5000 2>/dev/null | grep '^{' > gpurun_out/r6/loop2_resnet18_me.json
python tools/loop_bench.py --workload vgg19_me 2>/dev/null | grep '^{' > gpurun_out/r6/loop2_vgg19_me.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6/loop2_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["macro_batches"], d["pipe"], "loop", d["loop_mcd_samples_per_s"], "device", d["device_only_mcd_samples_per_s"], "overhead %", d["loop_overhead_pct"], d["all_loop_s"])
PY
B="--no-cpu-baseline --no-rccl-probe --no-parity-leg"
run() { tag=$1; opts=$2; shift; shift; BMI_OPTIONS="$opts" python bench.py $B "$@" 2>/dev/null | grep '^{' > gpurun_out/r6/${tag}.json; python - <<PY
import json
d=json.load(open("gpurun_out/r6/${tag}.json")); print("${tag}", d["value"], d["ms_per_step"], d["config"]["pipe"][:12], d["config"]["rank_step_probe_ms"], d["roofline"]["whole_step"]["frac"])
PY
}
for rep in 1 2; do
run y_exit_base_$rep "" --workload resnet18_exit_only
run y_exit_pairprefix_$rep "pair_prefix=1" --workload resnet18_exit_only
run y_exit_s2nomin_$rep "conv_s2=2" --workload resnet18_exit_only
run y_exit_pair_s2_$rep "pair_prefix=1,conv_s2=2" --workload resnet18_exit_only
run y_vgg19_base_$rep "" --workload vgg19_me
run y_vgg19_s2nomin_$rep "conv_s2=2" --workload vgg19_me
done
echo done
