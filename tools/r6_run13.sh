#!/bin/bash
# round-6 GPU run 13: full GPU suite + smoke on the final code; profiles of the two workloads whose 16x16 tails changed
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q --maxfail=40 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_13.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_13.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/profile_all.sh r06 resnet18_masksembles resnet18_me > /dev/null 2>&1
python - <<'PY'
import json
for W in ("resnet18_me","resnet18_masksembles"):
    d=json.load(open(f"gpurun_out/r06/r06_{W}_bench_line.json")); r=d["roofline"]
    print(W, d["value"], d["ms_per_step"], r["whole_step"]["frac"], r["kernel"], r["frac"], r.get("frac_rocprof"), d.get("value_at_tolerance"))
PY
echo done
