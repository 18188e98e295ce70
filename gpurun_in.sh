timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_extra_models.py tests/test_vgg.py -x -q -m gpu -k "dense or vgg or extra or VGG" 2>&1 | tail -3
timeout 600 python tools/per_launch.py --workload vgg11 2>&1 | grep -v amdgpu.ids | grep "dense\|launches"
timeout 300 python bench.py --workload vgg11 --no-cpu-baseline 2>/dev/null | cut -c1-200
