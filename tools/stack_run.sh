#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -f $O/stack_215.txt
timeout -k 10 500 python tools/stack_bench.py --lap 215 --dbg --reps 12 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --lap 215 --dtype s --reps 12 >> $O/stack_215.txt 2>&1 && \
timeout -k 10 500 python tools/stack_bench.py --band 3 --reps 12 >> $O/stack_215.txt 2>&1 && \
cat $O/stack_215.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stack_stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --ilu-side 0 --solve-side 0 > $O/stack_stats.log 2>&1 || exit 2
python3 - <<'PY'
import csv, glob, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/stack_stats'
f=glob.glob(O+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%-70s calls=%5s avg=%9.1f us  %6s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
