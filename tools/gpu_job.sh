#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/experiments/soak_masked_s2.py 400 2>&1 | tail -2
