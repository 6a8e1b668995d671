# round 3, run 3: v3.1 kernel (early loads, block level, readlane fine tests, group prefetch): exactness, A/B, splits for C5, phase stats
set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_apd_gpu.py tests/test_configs_gpu.py tests/test_c5_gpu.py tests/test_real_clouds.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03/t3.log 2>&1
rc=$?
tail -15 gpurun_out/r03/t3.log
[ $rc -eq 0 ] || exit $rc
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
for v in nn_s12w4 nn_s8w5 nn_s8w4; do
  GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab3_$v.json 2> gpurun_out/r03/ab3_$v.err || exit 1
done
for v in nn_s12w4 nn_s8w5; do
 for sp in 1 2 4; do
  GORIO_NN_SPLITS=$sp GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c53_${v}_sp$sp.json 2> gpurun_out/r03/c53_${v}_sp$sp.err || exit 1
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab3_*.json')+glob.glob('gpurun_out/r03/c53_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w4.so timeout -k 10 300 python tools/search_work.py c4 20 > gpurun_out/r03/stats3_c4.txt 2>&1 || { tail -20 gpurun_out/r03/stats3_c4.txt; exit 1; }
cat gpurun_out/r03/stats3_c4.txt
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats_s8w4.so timeout -k 10 400 python tools/search_work.py c5 20 > gpurun_out/r03/stats3_c5.txt 2>&1 || { tail -20 gpurun_out/r03/stats3_c5.txt; exit 1; }
cat gpurun_out/r03/stats3_c5.txt
