timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "head" 2>&1 | tail -3
timeout 500 python tools/head_bench.py 2>&1 | grep -v amdgpu
timeout 500 python tools/head_bench.py --cases 100:8,100:100 --K 2048 2>&1 | grep -v amdgpu
