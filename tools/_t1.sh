cd $GRAFT_REPO_ROOT
for dbg in 0 8; do for sch in 1 0; do echo "== fe debug=$dbg sched=$sch"; RLH_SPMM_SCHED=$sch RLH_WIDE_DEBUG=$dbg timeout -k 10 100 python tools/microbench.py --fe --m 16 --only spmm 2>&1 | grep "per application"; done; done
echo "== band15 nt"; RLH_WIDE_DEBUG=8 timeout -k 10 200 python tools/microbench.py --n 9938375 --m 32 --band 15 --only spmm 2>&1 | grep "spmm band"
echo "== band15"; timeout -k 10 200 python tools/microbench.py --n 9938375 --m 32 --band 15 --only spmm 2>&1 | grep "spmm band"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_all1.log 2>&1; echo rc=$?; tail -5 gpurun_out/t_all1.log
