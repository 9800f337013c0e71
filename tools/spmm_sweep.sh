#!/bin/bash
# Tuning sweep of the SLICED-ELL SpMM kernel on the GPU box (the windowed kernel has its own
# tunables: RLH_SPMM_CPS, RLH_SPMM_VEC, RLH_SPMM_SCHED; see tools/pmc_well.sh).
for jt in ${JTS:-32 16}; do for tt in ${TTS:-1 4 8}; do for bpc in ${BPCS:-0 1 2}; do
  echo "== JT=$jt TT=$tt WG_PER_CU=$bpc"; RLH_SPMM_FORMAT=sell RLH_SPMM_JT=$jt RLH_SPMM_TT=$tt RLH_SPMM_WG_PER_CU=$bpc RLH_SPMM_CHUNK=${CH:-64} python tools/microbench.py --lap ${LAP:-215} --m ${M:-32} --dtype d --only spmm 2>&1 | grep spmm
done; done; done
