#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 32 24; do timeout -k 10 120 python tools/gram_series.py $m || exit 1; done
echo "== waves 8"; RLH_GRAM_STREAM_WAVES=8 timeout -k 10 120 python tools/gram_series.py 32 || exit 1
echo "== waves 3"; RLH_GRAM_STREAM_WAVES=3 timeout -k 10 120 python tools/gram_series.py 32 || exit 1
