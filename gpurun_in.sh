timeout 1500 python -m pytest tests/test_vgg.py tests/test_extra_models.py tests/test_gpu_model.py tests/test_full_batch.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/per_launch.py --workload vgg11 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/per_launch.py --workload vgg11 --set splitk=0 2>&1 | grep -v amdgpu.ids | head -3
timeout 300 python bench.py --workload vgg11 --no-cpu-baseline 2>/dev/null | cut -c1-300
timeout 300 python bench.py --workload vgg19_me --no-cpu-baseline 2>/dev/null | cut -c1-300
