#!/bin/bash
# round-6 GPU run 8: new own-size parity tests + VGG FullAnalysis fix, loader walks after the device-side macro concat, then the r06 profiles (part 1)
mkdir -p gpurun_out/r6
python -m pytest tests/test_auto_engine.py tests/test_vgg.py tests/test_multi_gpu_mirrors.py -m gpu -q -s --maxfail=30 -rf -p no:cacheprovider > gpurun_out/r6/gpu_tests_8.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r6/gpu_tests_8.log; grep "auto ->" gpurun_out/r6/gpu_tests_8.log
for K in 1 4 8; do python tools/loop_bench.py --workload resnet18_exit_only --macro $K 2>/dev/null | grep '^{' > gpurun_out/r6/loop3_exit_only_macro$K.json; done
python tools/loop_bench.py --workload vgg19_me 2>/dev/null | grep '^{' > gpurun_out/r6/loop3_vgg19_me.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6/loop3_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["macro_batches"], d["pipe"], "loop", d["loop_mcd_samples_per_s"], "device", d["device_only_mcd_samples_per_s"], "overhead %", d["loop_overhead_pct"], d["all_loop_s"])
PY
tools/profile_all.sh r06 resnet18_me resnet18_exit_only vgg19_me resnet18_layer
echo done
