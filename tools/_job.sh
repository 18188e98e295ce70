python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^B+E\|^Ensemble" | tail -6
