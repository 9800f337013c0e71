#!/bin/bash
# GPU box: the checks and measurements of the stacked SpMM / bfloat16 Chebyshev kernels in one call
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/stack_run.sh'
# tests of the stacked paths, two seeds of the randomised sweep, the layouts taking turns on the headline operator and on
# config 5's, the bfloat16 step against the unstacked kernel at full size, and the 10^7-row solve with and without it.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "stack or bf16 or cheb" 2>&1 | tail -2 || exit 1
timeout -k 10 900 python tools/fuzz_spmm_stack.py 150 1 2>&1 | tail -1 | cut -c1-250
timeout -k 10 900 python tools/fuzz_spmm_stack.py 150 2 2>&1 | tail -1 | cut -c1-250
timeout -k 10 500 python tools/stack_bench.py --lap 215 --reps 12 2>&1 | cut -c1-150
timeout -k 10 500 python tools/stack_bench.py --herm 126 --dtype z --m 64 --reps 12 2>&1 | cut -c1-150
timeout -k 10 300 python tools/bf16_check.py 215 16 2>&1 | grep "finite\|vectors\|row blocks" | cut -c1-200
timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 2>&1 | grep "status 0\|Error" | cut -c1-200
RLH_SPMM_STACK_BF16=0 timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 2>&1 | grep "status 0\|Error" | cut -c1-200
