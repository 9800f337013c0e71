#!/bin/bash
# Round-3 profile set on the GPU box (rocprofv3; counters in their own passes, as MI355X_MICROARCH.md prescribes):
#  1. tools/profile_round.sh: the bench line, kernel stats of the headline command, FETCH_SIZE / WRITE_SIZE of its kernels
#  2. this round's kernels: the persistent triangular solve (three operators), the 16x16x4 matrix-core dense apply in
#     double / complex double (kernel stats + MFMA-busy counters), the forced-collectives bench with the sharded config-4 leg
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/round.txt 2>&1 || { tail -5 $O/round.txt; exit 1; }
tail -42 $O/round.txt
cd $R && timeout -k 10 400 python bench.py --gpus 1 --force-dist --no-cpu-baseline > $O/bench_forced.json 2> $O/bench_forced.err || echo "FAILED forced bench"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters ('' = kernel trace + stats), command...
  local name=$1 pmc=$2; shift 2
  if [ -z "$pmc" ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  else
    timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  fi
}
run ilu_fe_stats "" python3 $R/tools/ilu_bench.py fe
run ilu_lap100_stats "" python3 $R/tools/ilu_bench.py lap100
run si_lap30_stats "" python3 $R/tools/si_bench.py 30 --m 8
run gemm_d_stats "" python3 $R/tools/gemm_shapes.py --dtype d 20000x20000
run gemm_z_stats "" python3 $R/tools/gemm_shapes.py --dtype z 20000x20000
run gemm_d_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python3 $R/tools/gemm_shapes.py --dtype d 20000x20000
run solve_stats "" python3 $R/tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low --bf16
run stack_fetch "FETCH_SIZE" python3 $R/tools/stack_bench.py --lap 215 --reps 2
(cd $R && timeout -k 10 300 python3 tools/stack_bench.py --lap 215 --reps 12 > $O/stack_ab.txt 2>&1; timeout -k 10 300 python3 tools/stack_bench.py --herm 126 --dtype z --m 64 --reps 12 >> $O/stack_ab.txt 2>&1; timeout -k 10 300 python3 tools/pca_update_bench.py >> $O/stack_ab.txt 2>&1) || echo "FAILED stack_ab"
python3 - <<PY
import csv, glob, collections, os
O = "$O"
def newest(p):
    g = glob.glob(p)
    return max(g, key=os.path.getmtime) if g else None
for name in ("ilu_fe_stats", "ilu_lap100_stats", "si_lap30_stats", "gemm_d_stats", "gemm_z_stats", "solve_stats"):
    f = newest(O + "/%s/*/*kernel_stats.csv" % name)
    if not f: continue
    print("== %s (rocprofv3 --kernel-trace --stats)" % name)
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-70s calls=%5s avg=%10.1f us  %5s%%" % (r["Name"].split("(")[0].replace("void rlh::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
f = newest(O + "/stack_fetch/*/*counter_collection.csv")
if f:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "well_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void rlh::", "")[:60]].append(float(r["Counter_Value"]))
    print("== stack_fetch (rocprofv3 --pmc FETCH_SIZE, KiB; x 2 on gfx950 for wide coalesced reads)")
    for k, v in acc.items():
        print("  %-60s FETCH_SIZE %.0f KiB -> %.2f GB read over the fabric" % (k, sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e9))
try:
    print("== stack_ab (tools/stack_bench.py, tools/pca_update_bench.py)")
    print(open(O + "/stack_ab.txt").read())
except OSError:
    pass
for name in ("gemm_d_mfma",):
    f = newest(O + "/%s/*/*counter_collection.csv" % name)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void rlh::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s" % name)
    for k, v in acc.items():
        d = {c: sum(x) / len(x) for c, x in v.items()}
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
            print("  %-60s MFMA pipes busy %.1f %%  (%s)" % (k, 100 * busy, {c: int(x) for c, x in d.items()}))
PY
