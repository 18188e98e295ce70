#!/bin/bash
# round-6 GPU run 11: the full GPU suite + smoke + default bench on the final code, profiles refreshed for the workloads whose kernels changed since run 9
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q --maxfail=40 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_11.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_11.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
tools/profile_all.sh r06 resnet18_layer resnet18_exit_only vgg19_me resnet18_me
python - <<'PY'
import json
for W in ("resnet18_me","resnet18_exit_only","vgg19_me","resnet18_layer"):
    d=json.load(open(f"gpurun_out/r06/r06_{W}_bench_line.json")); r=d["roofline"]
    print(W, d["value"], d["ms_per_step"], r["whole_step"]["frac"], r["kernel"], r["frac"], r.get("frac_rocprof"))
PY
echo done
