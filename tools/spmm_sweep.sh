#!/bin/bash
# SpMM tuning sweep on the GPU box.
for jt in ${JTS:-32 16 8}; do for tt in ${TTS:-1}; do for ch in ${CHS:-64}; do for tpb in ${TPBS:-1 2 4}; do
  echo "== JT=$jt TT=$tt CHUNK=$ch TPB=$tpb"; RLH_SPMM_JT=$jt RLH_SPMM_TT=$tt RLH_SPMM_CHUNK=$ch RLH_SPMM_TPB=$tpb python tools/microbench.py --lap ${LAP:-215} --m ${M:-32} --dtype d --only spmm 2>&1 | grep spmm
done; done; done; done
