#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "stack or bf16" > $O/stack_tests.txt 2>&1 || { tail -40 $O/stack_tests.txt | cut -c1-200; exit 1; }
tail -2 $O/stack_tests.txt
timeout -k 10 500 python tools/stack_bench.py --lap 215 --dbg --reps 12 2>&1 | grep -v "timing-only" | cut -c1-150
timeout -k 10 500 python tools/stack_bench.py --lap 215 --dtype s --reps 12 2>&1 | cut -c1-150
timeout -k 10 500 python tools/stack_bench.py --herm 126 --dtype z --m 64 --reps 12 2>&1 | cut -c1-150
timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 2>&1 | grep "status 0\|Error" | cut -c1-200
