#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tier.txt 2>&1 || { tail -40 $O/gpu_tier.txt | cut -c1-200; exit 1; }
tail -3 $O/gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
