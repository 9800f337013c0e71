#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_driver_gpu.py -x -q 2>&1 | tail -3 || exit 1
for n in 27000 1000000 2000000; do echo "== n=$n m=32"; timeout -k 10 200 python tools/microbench.py --n $n --m 32 --only "gram" | tail -2 || exit 1; done
echo "== n=2M stream off"; RLH_GRAM_STREAM=0 timeout -k 10 200 python tools/microbench.py --n 2000000 --m 32 --only "gram" | tail -2
echo "== n=1M stream forced"; RLH_GRAM_STREAM=2 timeout -k 10 200 python tools/microbench.py --n 1000000 --m 32 --only "gram" | tail -2
