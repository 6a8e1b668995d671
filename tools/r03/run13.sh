set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_apd_gpu.py tests/test_real_clouds.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03/t13.log 2>&1
rc=$?
tail -3 gpurun_out/r03/t13.log
[ $rc -eq 0 ] || exit $rc
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab13_knn64.json 2> gpurun_out/r03/ab13_knn64.err || exit 1
GORIO_AMD_LIB=$PWD/tools/variants/knn_b256.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab13_knn256.json 2> gpurun_out/r03/ab13_knn256.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exhaustive > gpurun_out/r03/ab13_overlap.json 2> gpurun_out/r03/ab13_overlap.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab13_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
