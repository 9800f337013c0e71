#!/bin/bash
# Round-2 starting point on the GPU box: per-kernel numbers of the configurations VERDICT r01 lists as
# unmeasured (config-3 surrogate SpMM, config-5 complex128 m = 64 ops, band-15 SpMM, block update).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_base; mkdir -p $O
cd $R
timeout -k 10 200 python tools/microbench.py --fe --m 16 --only spmm > $O/fe_m16.txt 2>&1 && cat $O/fe_m16.txt
timeout -k 10 300 python tools/microbench.py --herm 126 --dtype z --m 64 > $O/herm126_z64.txt 2>&1 && cat $O/herm126_z64.txt
timeout -k 10 300 python tools/microbench.py --n 9938375 --m 32 --band 15 --only spmm > $O/band15.txt 2>&1 && cat $O/band15.txt
timeout -k 10 300 python tools/microbench.py --n 9938375 --m 32 > $O/d32.txt 2>&1 && cat $O/d32.txt
