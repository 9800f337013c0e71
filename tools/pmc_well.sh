#!/bin/bash
# PMC passes for the windowed SpMM kernel (one counter set per run, kernel-trace off).
# CASES: "<tag> <env assignments> -- <microbench args>"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_well; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run_case() {
 tag=$1; shift
 i=0
 for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_WRREQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/${tag}_s$i -- python $R/tools/microbench.py "$@" --m 32 --dtype d --only spmm --reps 3 > $O/${tag}_s$i.log 2>&1 || exit 1
 done
}
run_case lap3d --lap 215
RLH_SPMM_SCHED=0 run_case lap3d_nosched --lap 215
run_case band3 --n 9938375 --band 3
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_well'
for d in sorted(glob.glob(O+'/*_s*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'spmm' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
