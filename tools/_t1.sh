cd $GRAFT_REPO_ROOT
for ct in 0 512 2048; do for sl in 2 8; do echo "chain_tasks=$ct sl=$sl"; RLH_SPTRSV_CHAIN_TASKS=$ct RLH_SPTRSV_SLICES=$sl timeout -k 10 300 python tools/ilu_bench.py fe 2>&1 | grep "apply m"; done; done
for ct in 0 512 4096; do echo "lap30 chain_tasks=$ct"; RLH_SPTRSV_CHAIN_TASKS=$ct timeout -k 10 300 python tools/ilu_bench.py lap30 2>&1 | grep "apply m"; done
for ct in 0 4096 65536; do echo "lap100 chain_tasks=$ct"; RLH_SPTRSV_CHAIN_TASKS=$ct timeout -k 10 300 python tools/ilu_bench.py lap100 2>&1 | grep "apply m"; done
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_driver_gpu.py -x -q -m gpu > gpurun_out/t_cfg3.log 2>&1; echo rc=$?; tail -4 gpurun_out/t_cfg3.log
