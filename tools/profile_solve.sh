#!/bin/bash
# Kernel-time breakdown of the end-to-end solve (10 eigenpairs of lap3d 215^3, float32 Chebyshev
# preconditioner): rocprofv3 --kernel-trace --stats around tools/solve_lap.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/solve_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/solve_lap.py --side ${SIDE:-215} --cheb 32 --ratio 7000 --low ${EXTRA:---bf16} > $O/solve.log 2>&1 || { tail -5 $O/solve.log; exit 2; }
tail -3 $O/solve.log
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/solve_prof'
f = glob.glob(O + '/stats/*/*kernel_stats.csv')[0]
tot = 0
rows = list(csv.DictReader(open(f)))
for r in rows:
    tot += float(r['TotalDurationNs'])
print('total kernel time %.1f ms' % (tot / 1e6))
for r in rows[:14]:
    print('%-70s calls=%5s avg=%9.1f us total=%8.1f ms %5s%%' % (r['Name'].replace('void rlh::', '')[:70], r['Calls'],
          float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
