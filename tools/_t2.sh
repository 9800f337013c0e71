#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_hip_parity_gpu.py -x -q -k "triang or ilu or chain or shift or sptrsv or config" 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/small_solve.py 2>&1 | tail -5
timeout -k 10 300 python tools/ilu_bench.py fe | tail -3
timeout -k 10 300 python tools/ilu_bench.py lap100 | tail -3
