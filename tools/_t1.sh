cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "spmm or layout or cheb or config3" > gpurun_out/t_spmm5.log 2>&1; echo rc=$? ; tail -3 gpurun_out/t_spmm5.log
for nv in 8 16; do echo "== fe NV=$nv"; RLH_WIDE_NV=$nv timeout -k 10 100 python tools/microbench.py --fe --m 16 --only spmm 2>&1 | grep -v "^n="; done
for dbg in 1 2 3 7; do echo "== fe NV=16 debug=$dbg"; RLH_WIDE_DEBUG=$dbg RLH_WIDE_NV=16 timeout -k 10 100 python tools/microbench.py --fe --m 16 --only spmm 2>&1 | grep "per application"; done
for nv in 16 32; do echo "== band15 NV=$nv"; RLH_WIDE_NV=$nv timeout -k 10 200 python tools/microbench.py --n 9938375 --m 32 --band 15 --only spmm 2>&1 | grep -v "^n="; done
for nv in 4 8; do echo "== herm z NV=$nv"; RLH_WIDE_NV=$nv timeout -k 10 200 python tools/microbench.py --herm 126 --dtype z --m 64 --only spmm 2>&1 | grep -v "^n="; done
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s > gpurun_out/t_configs2.log 2>&1; echo rc=$? ; tail -8 gpurun_out/t_configs2.log
