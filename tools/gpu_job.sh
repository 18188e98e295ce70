#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_sc1
mkdir -p $O
timeout 900 python3 -m pytest tests/test_converter.py tests/test_gpu_model.py tests/test_sharding.py -q > $O/test_new.log 2>&1; echo "new tests rc=$?" >> $O/test_new.log
tools/ab_any.sh "python3 tools/conv_bench.py --images 12000 --iters 10 --rounds 3 --nores --sparse-input --only D2p,D3,D3p,D4,D4p" base sc1 > $O/ab.log 2>&1
cp bayesnn_fpga_amd/csrc/build/variants/lib_sc1.so /tmp/lib_sc1.so; cp bayesnn_fpga_amd/libbayesnn_fpga_amd.so /tmp/lib_base.so
cp /tmp/lib_sc1.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_sc1.log 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 $R/bench.py --workload resnet18_me --steps 1 --warmup 1 --no-cpu-baseline --in-flight 1 > $O/fetch.log 2>&1
cd $R
cp /tmp/lib_base.so bayesnn_fpga_amd/libbayesnn_fpga_amd.so
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_base.log 2>&1
python3 - <<'PY' > $O/fetch_per_dispatch.txt 2>&1
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r3_sc1/fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE" and ("conv3x3_s2" in row["Kernel_Name"]):
            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"][:60], float(row["Counter_Value"]) * 1024 * 2 / 1e9))
for r in sorted(rows)[-6:]:
    print(r[0], r[1], round(r[2], 3), "GB")
PY
rm -rf $O/fetch
tail -4 $O/test_new.log; grep "==\|median" $O/ab.log | cut -c1-100
for f in bench_base bench_sc1; do echo $f; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/$f.log | tr '\n' ' '; echo; done
cat $O/fetch_per_dispatch.txt
