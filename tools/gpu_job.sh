#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_full_batch.py -x -q -k "bf16" 2>&1 | tail -8
