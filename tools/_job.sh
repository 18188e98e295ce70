python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "head or dense" 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py tests/test_converter.py tests/test_vgg.py -m gpu -x -q 2>&1 | tail -3
python tools/step_ab.py --rounds 5 --steps 3 --ab "xcd_split=0" 2>&1 | grep -v amdgpu
python tools/step_ab.py --T 13 --rounds 5 --steps 5 --ab "xcd_split=0" 2>&1 | grep -v amdgpu
python tools/step_ab.py --workload resnet18_masksembles --rounds 3 --steps 3 --ab "xcd_split=0" 2>&1 | grep -v amdgpu
