#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -k "update or combine2" 2>&1 | tail -5
