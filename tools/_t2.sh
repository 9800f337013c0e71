#!/bin/bash
cd $GRAFT_REPO_ROOT
for rv in 0 1; do echo "== RV64=$rv"; RLH_UPDATE_RV64=$rv timeout -k 10 300 python tools/update2_bench.py; done
