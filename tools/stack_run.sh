#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -f $O/stack_215.txt
timeout -k 10 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "spmm or sparse or cheb or halo or sharded" > $O/stack_tests.txt 2>&1 || { tail -40 $O/stack_tests.txt | cut -c1-200; exit 1; }
tail -3 $O/stack_tests.txt
timeout -k 10 600 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-configs > $O/bench_fd_stack.json 2> $O/bench_fd_stack.err || { tail -5 $O/bench_fd_stack.err; exit 1; }
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/bench_fd_stack.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['inner_iteration_frac'], d.get('collectives'))
PY
