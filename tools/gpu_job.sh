#!/bin/bash
mkdir -p gpurun_out/r3_kw
cd $GRAFT_REPO_ROOT
for KW in "10 3" "10 10" "30 3" "60 5" "5 1"; do set -- $KW
echo -n "steps $1 warmup $2: "; python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done 2>&1 | tee gpurun_out/r3_kw/kw.log
