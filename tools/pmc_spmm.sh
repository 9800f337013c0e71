#!/bin/bash
# PMC passes for the sliced-ELL SpMM kernel (one counter set per run, kernel-trace off).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_spmm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for jt in ${JTS:-32 16 4}; do
 i=0
 for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  RLH_SPMM_FORMAT=sell RLH_SPMM_JT=$jt rocprofv3 --pmc $set --output-format csv -d $O/jt${jt}_s$i -- python $R/tools/microbench.py --lap 215 --m 32 --dtype d --only spmm --reps 3 > $O/jt${jt}_s$i.log 2>&1 || exit 1
 done
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_spmm'
for d in sorted(glob.glob(O+'/jt*_s*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'spmm' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
