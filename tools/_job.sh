python -m pytest tests/test_dynamic_exit.py -m gpu -x -q -s 2>&1 | tail -15
