#!/bin/bash
mkdir -p gpurun_out/r3_final
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_full_batch.py -x -q -k "lazy" -s 2>&1 | tail -6
