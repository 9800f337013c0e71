cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_d; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 $GRAFT_REPO_ROOT/tools/gemm_shapes.py --dtype d 20000x20000 > $O/a.log 2>&1 || echo FAILED
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/a/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    d = {c: sum(x) / len(x) for c, x in v.items()}
    if d.get("SQ_WAVE_CYCLES", 0) > 1e6:
        print(k)
        for c, x in sorted(d.items()): print("   %-28s %14.0f  (%.3f of wave cycles)" % (c, x, x / d["SQ_WAVE_CYCLES"]))
PY
