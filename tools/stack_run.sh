#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for seed in 2 3 5 6; do
timeout -k 10 900 python tools/fuzz_spmm_stack.py 150 $seed > $O/fuzz_stack_$seed.txt 2>&1; tail -1 $O/fuzz_stack_$seed.txt | cut -c1-300
done
