#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_dist_gpu.py tests/test_driver_gpu.py -x -q -m gpu -k "bf16 or cheb or sharded or driver" > $O/stack_tests.txt 2>&1 || { tail -40 $O/stack_tests.txt | cut -c1-200; exit 1; }
tail -3 $O/stack_tests.txt
RLH_SPMM_BF16_VPS=1 timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "bf16_on_stacks" 2>&1 | tail -2
timeout -k 10 300 python tools/bf16_check.py 215 16 2>&1 | grep "finite\|vectors\|row blocks" | cut -c1-200
timeout -k 10 300 python tools/bf16_check.py 215 13 2>&1 | grep "finite\|vectors\|row blocks" | cut -c1-200
timeout -k 10 600 python tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16 2>&1 | grep "status 0\|Error" | cut -c1-200
