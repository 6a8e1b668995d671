set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_apd_gpu.py tests/test_configs_gpu.py tests/test_c5_gpu.py tests/test_real_clouds.py tests/test_golden.py tests/test_submap_gpu.py -m gpu -x -q > gpurun_out/r03/t12.log 2>&1
rc=$?
tail -3 gpurun_out/r03/t12.log
[ $rc -eq 0 ] || exit $rc
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab12_plan.json 2> gpurun_out/r03/ab12_plan.err || exit 1
GORIO_NN_NO_PLAN=1 timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab12_noplan.json 2> gpurun_out/r03/ab12_noplan.err || exit 1
GORIO_AMD_LIB=$PWD/tools/variants/nn_w5.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab12_w5_plan.json 2> gpurun_out/r03/ab12_w5_plan.err || exit 1
timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c512_plan.json 2> gpurun_out/r03/c512_plan.err || exit 1
GORIO_AMD_LIB=$PWD/tools/variants/nn_w5.so timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c512_w5_plan.json 2> gpurun_out/r03/c512_w5_plan.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab12_*.json')+glob.glob('gpurun_out/r03/c512_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
GORIO_AMD_LIB=$PWD/tools/variants/nn_stats.so timeout -k 10 300 python tools/search_work.py c4 20 2>&1 | grep -v "cycles per wave\|share"
