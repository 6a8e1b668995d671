set -x
mkdir -p gpurun_out/r03
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w5.so timeout -k 10 300 python tools/search_work.py c4 20 > gpurun_out/r03/stats_c4.txt 2>&1 || { tail -20 gpurun_out/r03/stats_c4.txt; exit 1; }
cat gpurun_out/r03/stats_c4.txt
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w5.so timeout -k 10 400 python tools/search_work.py c5 20 > gpurun_out/r03/stats_c5.txt 2>&1 || { tail -20 gpurun_out/r03/stats_c5.txt; exit 1; }
cat gpurun_out/r03/stats_c5.txt
