set -x
timeout 500 python tools/experiments/lite_vs_general.py 2>&1 | tail -13
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
