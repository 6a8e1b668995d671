set -x
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w5b64.so timeout -k 10 300 python tools/search_work.py c4 20 > gpurun_out/r03/stats7_c4.txt 2>&1 || { tail -20 gpurun_out/r03/stats7_c4.txt; exit 1; }
cat gpurun_out/r03/stats7_c4.txt
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w5b64.so timeout -k 10 400 python tools/search_work.py c5 20 > gpurun_out/r03/stats7_c5.txt 2>&1 || { tail -20 gpurun_out/r03/stats7_c5.txt; exit 1; }
cat gpurun_out/r03/stats7_c5.txt
