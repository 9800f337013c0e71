#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_pca_update_gpu.py tests/test_pca_complex_gpu.py -x -q -m gpu > $O/pca_tests.txt 2>&1 || { tail -30 $O/pca_tests.txt | cut -c1-200; exit 1; }
tail -3 $O/pca_tests.txt
PCA_UPDATE_PROFILE=1 timeout -k 10 600 python tools/pca_update_bench.py > $O/pca_update_now.txt 2>&1 || { tail -20 $O/pca_update_now.txt; exit 1; }
head -22 $O/pca_update_now.txt | cut -c1-170
RLH_DEVICE_EIGH=0 timeout -k 10 600 python tools/pca_update_bench.py > $O/pca_update_host.txt 2>&1 || { tail -20 $O/pca_update_host.txt; exit 1; }
head -4 $O/pca_update_host.txt | cut -c1-170
