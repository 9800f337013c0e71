cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "padding" > gpurun_out/t_spmm2.log 2>&1; echo rc=$? ; tail -3 gpurun_out/t_spmm2.log
for nv in 8 16; do for wg in 1 2 4; do echo "== fe NV=$nv WG=$wg"; RLH_WIDE_NV=$nv RLH_WIDE_WG_PER_CU=$wg timeout -k 10 100 python tools/microbench.py --fe --m 16 --only spmm 2>&1 | grep -v "^n="; done; done
for nv in 16 32; do for wg in 1 2 4; do echo "== band15 NV=$nv WG=$wg"; RLH_WIDE_NV=$nv RLH_WIDE_WG_PER_CU=$wg timeout -k 10 200 python tools/microbench.py --n 9938375 --m 32 --band 15 --only spmm 2>&1 | grep -v "^n="; done; done
for nv in 4 8 16; do echo "== herm z NV=$nv"; RLH_WIDE_NV=$nv timeout -k 10 200 python tools/microbench.py --herm 126 --dtype z --m 64 --only spmm 2>&1 | grep -v "^n="; done
echo "== lap3d 215 wide"; RLH_SPMM_FORMAT=wide timeout -k 10 200 python tools/microbench.py --lap 215 --m 32 --only spmm 2>&1 | grep -v "^n="
echo "== lap3d 215 well"; timeout -k 10 200 python tools/microbench.py --lap 215 --m 32 --only spmm 2>&1 | grep -v "^n="
