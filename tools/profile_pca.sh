#!/bin/bash
# PCA path (BASELINE config 2) on the GPU box: dense-apply TFLOP/s + rocprofv3 kernel stats.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && timeout -k 10 400 python tools/pca_bench.py --gemm-only > $O/pca_gemm.txt 2>&1 || exit 1
cat $O/pca_gemm.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pca_stats -- python $R/tools/pca_bench.py --gemm-only > $O/pca_stats.log 2>&1 || exit 2
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pca_pmc -- python $R/tools/pca_bench.py --gemm-only > $O/pca_pmc.log 2>&1 || exit 3
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/pca_stats/*/*kernel_stats.csv")[0]
print("== rocprofv3 --kernel-trace --stats (tools/pca_bench.py --gemm-only, 20000 x 20000 fp32, m = 128)")
for r in csv.DictReader(open(f)):
    print("%-64s calls=%4s avg=%10.1f us" % (r["Name"].split("(")[0].replace("void rlh::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
f = glob.glob("$O/pca_pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "dense_mfma" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC per dense_mfma_f32_kernel launch (avg):", {k: sum(v) / len(v) for k, v in acc.items()})
if "GRBM_GUI_ACTIVE" in acc:
    print("GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles per launch" % (sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8))
PY
