# round 3, run 1: exactness of the rewritten pruned 1-NN kernel, then A/B against the round-2 kernel (GORIO_NN_V2) and slot/occupancy variants
set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_apd_gpu.py tests/test_configs_gpu.py tests/test_c5_gpu.py tests/test_real_clouds.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03/t1.log 2>&1
rc=$?
tail -15 gpurun_out/r03/t1.log
[ $rc -eq 0 ] || exit $rc
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
GORIO_NN_V2=1 timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab_v2.json 2> gpurun_out/r03/ab_v2.err || exit 1
for v in nn_s12w4 nn_s8w5 nn_s16w3; do
  GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab_$v.json 2> gpurun_out/r03/ab_$v.err || exit 1
done
GORIO_NN_V2=1 timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c5_v2.json 2> gpurun_out/r03/c5_v2.err || exit 1
for v in nn_s12w4 nn_s8w5; do
  GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c5_$v.json 2> gpurun_out/r03/c5_$v.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab_*.json')+glob.glob('gpurun_out/r03/c5_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
