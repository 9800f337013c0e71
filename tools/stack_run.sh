#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -f $O/stack_215.txt
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "spmm or sparse or cheb" > $O/stack_tests.txt 2>&1 || { tail -30 $O/stack_tests.txt | cut -c1-200; exit 1; }
tail -3 $O/stack_tests.txt
RLH_SPMM_STACK=2 timeout -k 10 300 python tools/stack_bench.py --herm 42 --dtype z --check --reps 3 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --herm 126 --dtype z --m 64 --reps 12 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --herm 128 --dtype z --m 64 --reps 12 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --herm 160 --dtype c --m 64 --reps 12 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --lap 215 --reps 12 >> $O/stack_215.txt 2>&1 && \
cat $O/stack_215.txt
