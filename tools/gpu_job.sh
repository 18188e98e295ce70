#!/bin/bash
mkdir -p gpurun_out/r3_final
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 900 python -m pytest tests/test_conv3x3_s2.py tests/test_full_batch.py tests/test_dynamic_exit.py tests/test_race_screen.py -x -q 2>&1 | tail -1; done | tee gpurun_out/r3_final/repeat.log
for i in 1 2 3 4 5; do python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-mean gpurun_out/r3_final/mean_$i.npy 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
python3 -c "
import numpy as np
m=[np.load('gpurun_out/r3_final/mean_%d.npy'%i) for i in range(1,6)]
print('repeat runs bit-identical:', all(np.array_equal(m[0],x) for x in m[1:]), m[0].shape)
"
rm -f gpurun_out/r3_final/mean_*.npy
