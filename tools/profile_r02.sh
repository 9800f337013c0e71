#!/bin/bash
# Round-2 profile set on the GPU box (rocprofv3; counters in their own passes, as MI355X_MICROARCH.md prescribes):
#  1. tools/profile_round.sh: bench line, kernel stats of the headline, FETCH_SIZE / WRITE_SIZE of its kernels
#  2. the new kernels: interleaved windowed SpMM (config-3 surrogate, band of 31 entries), 128 x 128 Gram panels and
#     the matrix-core block update (config 5), level-scheduled triangular solves, fp32 dense apply (config 2)
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/round.txt 2>&1 || { tail -5 $O/round.txt; exit 1; }
tail -40 $O/round.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters ('' = kernel trace + stats), command...
  local name=$1 pmc=$2; shift 2
  if [ -z "$pmc" ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  else
    timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  fi
}
for what in fe band; do
  if [ $what = fe ]; then ARGS="--fe --m 16 --only spmm"; else ARGS="--n 9938375 --m 32 --band 15 --only spmm"; fi
  run spmm_${what}_stats "" python3 $R/tools/microbench.py $ARGS
  run spmm_${what}_fetch FETCH_SIZE python3 $R/tools/microbench.py $ARGS
  run spmm_${what}_write WRITE_SIZE python3 $R/tools/microbench.py $ARGS
done
run c5_stats "" python3 $R/tools/microbench.py --n 2000376 --m 64 --dtype z
run c5_fetch FETCH_SIZE python3 $R/tools/microbench.py --n 2000376 --m 64 --dtype z --only gram
run c5_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python3 $R/tools/microbench.py --n 2000376 --m 64 --dtype z --only "X"
# (the profiler segfaults inside hipGraphLaunch when the captured level launches are traced: the library falls back to
# plain launches when it sees rocprofv3's ROCP_TOOL_LIBRARIES)
run ilu_stats "" python3 $R/tools/ilu_bench.py lap100
run ilu_fe_stats "" python3 $R/tools/ilu_bench.py fe
run pca_stats "" python3 $R/tools/pca_bench.py --gemm-only
run pca_mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python3 $R/tools/pca_bench.py --gemm-only
python3 - <<PY
import csv, glob, collections, os
O = "$O"
def newest(p):
    g = glob.glob(p)
    return max(g, key=os.path.getmtime) if g else None
for name in ("spmm_fe_stats", "spmm_band_stats", "c5_stats", "ilu_stats", "ilu_fe_stats", "pca_stats"):
    f = newest(O + "/%s/*/*kernel_stats.csv" % name)
    if not f: continue
    print("== %s (rocprofv3 --kernel-trace --stats)" % name)
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  %-70s calls=%5s avg=%10.1f us  %5s%%" % (r["Name"].split("(")[0].replace("void rlh::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for name, mult in (("spmm_fe_fetch", 2), ("spmm_fe_write", 1), ("spmm_band_fetch", 2), ("spmm_band_write", 1), ("c5_fetch", 2)):
    f = newest(O + "/%s/*/*counter_collection.csv" % name)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void rlh::", "")[:60]].append(float(r["Counter_Value"]))
    print("== %s (KiB counter x %d = bytes; gfx950: FETCH_SIZE counts half of a wide coalesced read)" % (name, mult))
    for k, v in acc.items():
        if len(v) >= 3: print("  %-60s launches=%4d  avg %.4f GB" % (k, len(v), mult * sum(v) / len(v) * 1024 / 1e9))
for name in ("c5_mfma", "pca_mfma"):
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
