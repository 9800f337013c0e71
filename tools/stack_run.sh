#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 --profile > $O/solve_profile.txt 2>&1 || { tail -20 $O/solve_profile.txt; exit 1; }
head -70 $O/solve_profile.txt | cut -c1-160
tail -12 $O/solve_profile.txt | cut -c1-200
