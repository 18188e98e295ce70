timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/per_launch.py --workload resnet18_me 2>&1 | grep -v amdgpu.ids | grep "mask\|launches"
