#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, and the two HBM PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).
# usage: tools/profile_round.sh r01
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --ilu-side 0 --ilu-large-side 0 --solve-side 0 > $O/stats.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --ilu-side 0 --ilu-large-side 0 --solve-side 0 > $O/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --ilu-side 0 --ilu-large-side 0 --solve-side 0 > $O/write.log 2>&1 || exit 4
python3 $R/tools/summarize_profile.py $O > $O/summary.txt && cat $O/summary.txt
