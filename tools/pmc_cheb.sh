#!/bin/bash
# PMC passes over the end-to-end solve, reported for the fused Chebyshev kernel only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_cheb; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python $R/tools/solve_lap.py --side 215 --cheb 32 --ratio 7000 --low ${EXTRA:---bf16} > $O/s$i.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_cheb'
for d in sorted(glob.glob(O+'/s*')):
    if not os.path.isdir(d): continue
    f=max(glob.glob(d+'/*/*counter_collection.csv'), key=os.path.getmtime)
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'cheb' in r['Kernel_Name'] or ('well_spmm' in r['Kernel_Name'] and 'float' in r['Kernel_Name']):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'launches', len(next(iter(acc.values()), [])))
PY
