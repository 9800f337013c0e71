#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "stack or bf16 or cheb" 2>&1 | tail -2
timeout -k 10 900 python tools/fuzz_spmm_stack.py 150 8 2>&1 | tail -1 | cut -c1-250
timeout -k 10 900 python tools/fuzz_spmm_stack.py 150 9 2>&1 | tail -1 | cut -c1-250
for dpat in 1 0; do
echo "== position patterns $dpat"
RLH_SPMM_STACK_DPAT=$dpat timeout -k 10 500 python tools/stack_bench.py --lap 215 --reps 12 2>&1 | grep "n =\|stacks, LDS\|blocks  " | cut -c1-150
RLH_SPMM_STACK_DPAT=$dpat timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 2>&1 | grep "status 0\|Error" | cut -c1-200
done
