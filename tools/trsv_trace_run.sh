RLH_SPTRSV_TRACE=/tmp/t_l.bin timeout -k 10 200 python tools/ilu_bench.py lap100 --m 16 2>&1 | grep -E "ilu apply|levels" > gpurun_out/r03_trace4.txt
python tools/trsv_trace.py /tmp/t_l.bin >> gpurun_out/r03_trace4.txt 2>&1
RLH_SPTRSV_TRACE=/tmp/t_fe.bin timeout -k 10 200 python tools/ilu_bench.py fe --m 16 2>&1 | grep -E "ilu apply|levels" >> gpurun_out/r03_trace4.txt
python tools/trsv_trace.py /tmp/t_fe.bin >> gpurun_out/r03_trace4.txt 2>&1
cat gpurun_out/r03_trace4.txt
