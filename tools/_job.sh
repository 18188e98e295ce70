python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -s 2>&1 | grep -v "^B+E\|^Ensemble" | tail -12
python bench.py --workload vgg11 --steps 5 --warmup 2 --dtype bf16 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['dtype'], d['cpu_baseline'])"
python bench.py --steps 5 --warmup 2 --dtype bf16 --cpu-T 4 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['dtype'], d['roofline']['frac'], d['cpu_baseline'])"
