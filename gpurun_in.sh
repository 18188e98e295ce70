timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/step_ab.py --workload resnet50_me --rounds 5 --steps 2 --ab "conv_stream=0,conv_stream=1" 2>&1 | tail -4
