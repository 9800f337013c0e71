#!/bin/bash
# Round-4 profile set on the GPU box (rocprofv3; counters in their own passes, as MI355X_MICROARCH.md prescribes):
#  1. tools/profile_round.sh: the bench line, kernel stats of the headline command, FETCH_SIZE / WRITE_SIZE of its kernels
#  2. the forced-collectives bench at one rank
#  3. this round's kernels: the config-5 inexact shift-invert solve (block MINRES), the complex128 Gram on LDS-DMA staging,
#     the split-K dense apply in double / complex (kernel stats + MFMA-busy counters), the config-5 SpMM on plane-aligned stacks
#  4. host set-up times, halo exchange breakdown, the direct shift-invert (L D L^H against SuperLU), what one rank's row shard launches
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/round.txt 2>&1 || { tail -5 $O/round.txt; exit 1; }
tail -42 $O/round.txt
cd $R && timeout -k 10 400 python bench.py --gpus 1 --force-dist --no-cpu-baseline --ilu-side 0 --ilu-large-side 0 > $O/bench_forced.json 2> $O/bench_forced.err || echo "FAILED forced bench"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters ('' = kernel trace + stats), command...
  local name=$1 pmc=$2; shift 2
  if [ -z "$pmc" ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  else
    timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  fi
}
run c5_solve_stats "" python3 $R/tools/config5_solve.py 126 40 16 250 1e-8 1e-6
run zgram_stats "" python3 $R/tools/zgram_bench.py 126
run zgram_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python3 $R/tools/zgram_bench.py 126
run gemm_d_stats "" python3 $R/tools/gemm_shapes.py --dtype d 20000x20000
run gemm_z_stats "" python3 $R/tools/gemm_shapes.py --dtype z 20000x20000
run gemm_c_stats "" python3 $R/tools/gemm_shapes.py --dtype c 20000x20000
run gemm_d_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python3 $R/tools/gemm_shapes.py --dtype d 20000x20000
run c5_ops_stats "" python3 $R/tools/c5_ops.py 126 64
(cd $R && timeout -k 10 300 python3 tools/gemm_shapes.py --dtype d 20000x20000 62500x40000 > $O/text.txt 2>&1; timeout -k 10 200 python3 tools/gemm_shapes.py --dtype z 20000x20000 >> $O/text.txt 2>&1; timeout -k 10 200 python3 tools/gemm_shapes.py --dtype c 20000x20000 >> $O/text.txt 2>&1
 for z in "1 0" "1 1" "1 2" "0 0"; do set -- $z; RLH_GRAM_ZDMA=$1 RLH_GRAM_ZDBG=$2 timeout -k 10 100 python3 tools/zgram_bench.py 126 >> $O/text.txt 2>&1; done
 timeout -k 10 200 python3 tools/c5_ops.py 126 64 >> $O/text.txt 2>&1
 timeout -k 10 200 python3 tools/stack_bench.py --herm 126 --dtype z --m 64 --reps 12 >> $O/text.txt 2>&1
 timeout -k 10 200 python3 tools/stack_bench.py --lap 215 --reps 12 >> $O/text.txt 2>&1
 timeout -k 10 300 python3 tools/setup_bench.py 215 0 >> $O/text.txt 2>&1
 timeout -k 10 300 python3 tools/ilut_bench.py 1 4 16 >> $O/text.txt 2>&1
 timeout -k 10 300 python3 tools/halo_bench.py 215 32 2>&1 | grep -v "NCCL\|RCCL\|version\|Hostname\|amdgpu.ids\|Librccl\|^\[W" >> $O/text.txt
 timeout -k 10 300 python3 tools/config5_solve.py 126 40 16 250 1e-8 1e-6 >> $O/text.txt 2>&1
 timeout -k 10 200 python3 tools/si_bench.py 30 --m 8 >> $O/text.txt 2>&1
 timeout -k 10 200 python3 tools/si_bench.py 40 --m 16 >> $O/text.txt 2>&1
 timeout -k 10 200 python3 tools/si_bench.py 30 --m 64 --complex --method ldlt >> $O/text.txt 2>&1
 for s in "8 1" "4 1" "2 0"; do set -- $s; timeout -k 10 200 python3 tools/c5_shard_bench.py $1 $2 2>&1 | grep "^shard" >> $O/text.txt; timeout -k 10 200 python3 tools/lap_shard_bench.py $1 $2 2>&1 | grep "^shard" >> $O/text.txt; done) || echo "FAILED text"
python3 - <<PY
import csv, glob, collections, os
O = "$O"
def newest(p):
    g = glob.glob(p)
    return max(g, key=os.path.getmtime) if g else None
for name in ("c5_solve_stats", "zgram_stats", "gemm_d_stats", "gemm_z_stats", "gemm_c_stats", "c5_ops_stats"):
    f = newest(O + "/%s/*/*kernel_stats.csv" % name)
    if not f: continue
    print("== %s (rocprofv3 --kernel-trace --stats)" % name)
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-70s calls=%5s avg=%10.1f us  %5s%%" % (r["Name"].split("(")[0].replace("void rlh::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for name in ("zgram_mfma", "gemm_d_mfma"):
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
print("== text (tools/gemm_shapes.py, zgram_bench.py, c5_ops.py, stack_bench.py, setup_bench.py, ilut_bench.py, halo_bench.py, config5_solve.py, si_bench.py, c5_shard_bench.py, lap_shard_bench.py)")
print(open(O + "/text.txt").read())
PY
