set -x
mkdir -p gpurun_out/r03 profiles/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_14.log 2>&1
rc=$?
tail -5 gpurun_out/r03/gpu_tests_14.log
[ $rc -eq 0 ] || exit $rc
bash tools/r03/pmc.sh c4 > gpurun_out/r03/pmc_c4.log 2>&1 || { tail -20 gpurun_out/r03/pmc_c4.log; exit 1; }
bash tools/r03/pmc.sh c3 > gpurun_out/r03/pmc_c3.log 2>&1 || { tail -20 gpurun_out/r03/pmc_c3.log; exit 1; }
cp profiles/kernel_counters.json gpurun_out/r03/kernel_counters_after_c3.json
cp -r profiles/r03 gpurun_out/r03/profiles_r03
ls profiles/r03 profiles/r03/c4 | head -30
