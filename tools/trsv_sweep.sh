# Sensitivity of the persistent triangular solve to the number of resident workgroups and the poll pause (congestion theory:
# the speculative polls of units far ahead of the front queue in front of the front's own gathers)
O=gpurun_out/trsv_sweep.txt; : > $O
for wg in 1 2 3 4; do for nap in 1 2 4; do
  echo "=== WG_PER_CU=$wg NAP=$nap" >> $O
  RLH_SPTRSV_WG_PER_CU=$wg RLH_SPTRSV_NAP=$nap timeout -k 10 200 python tools/ilu_bench.py lap100 --m 16 2>&1 | grep -E "ilu apply" >> $O
  RLH_SPTRSV_WG_PER_CU=$wg RLH_SPTRSV_NAP=$nap timeout -k 10 200 python tools/ilu_bench.py fe --m 16 2>&1 | grep -E "ilu apply" >> $O
done; done
cat $O
