# GPU check of the persistent triangular solve: the chain tests, then the apply timings the review names
timeout -k 10 300 python -m pytest tests/test_configs_gpu.py -x -q -k "triangular_chain" > gpurun_out/r03_trsv_t.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r03_trsv_t.txt; tail -3 gpurun_out/r03_trsv_t.txt
grep -q "rc=0" gpurun_out/r03_trsv_t.txt || exit 1
: > gpurun_out/r03_trsv_b.txt
for blk in default 1 ; do
if [ $blk = default ]; then unset RLH_SPTRSV_BLOCK; else export RLH_SPTRSV_BLOCK=$blk; fi
echo "=== RLH_SPTRSV_BLOCK $blk" >> gpurun_out/r03_trsv_b.txt
timeout -k 10 200 python tools/ilu_bench.py lap30 --m 16 2>&1 | grep -E "ilu apply|levels" >> gpurun_out/r03_trsv_b.txt || exit 1
timeout -k 10 200 python tools/ilu_bench.py fe --m 16 2>&1 | grep -E "ilu apply|levels" >> gpurun_out/r03_trsv_b.txt || exit 1
timeout -k 10 200 python tools/ilu_bench.py lap100 --m 16 2>&1 | grep -E "ilu apply|levels" >> gpurun_out/r03_trsv_b.txt || exit 1
timeout -k 10 200 python tools/si_bench.py 30 --m 8 >> gpurun_out/r03_trsv_b.txt 2>&1 || exit 1
done
cat gpurun_out/r03_trsv_b.txt
