#!/bin/bash
# PMC profile of the fp32 dense apply (PCA path): where the MFMA pipe's idle time goes.
TAG=${1:-r02_gemm}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1 -- python $R/tools/pca_bench.py --gemm-only > $O/pmc1.log 2>&1 || exit 3
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python $R/tools/pca_bench.py --gemm-only > $O/pmc2.log 2>&1 || exit 4
python3 - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    f = glob.glob("$O/%s/*/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void rlh::", "")[:60]
        if "dense" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(list(v.values())[0]))
PY
