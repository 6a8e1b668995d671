set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_apd_gpu.py tests/test_real_clouds.py tests/test_golden.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r03/t18.log 2>&1
rc=$?
tail -3 gpurun_out/r03/t18.log
[ $rc -eq 0 ] || exit $rc
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab18_base.json 2> gpurun_out/r03/ab18_base.err || exit 1
GORIO_AMD_LIB=$PWD/tools/variants/collect_w4.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab18_cw4.json 2> gpurun_out/r03/ab18_cw4.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab18_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/trace18 -o t -- python3 bench.py --steps 5 --warmup 1 --no-overlap --no-cpu-baseline --no-check --no-exhaustive > /dev/null 2>&1
grep "knn_\|kd_refine\|bitonic" gpurun_out/r03/trace18/t_kernel_stats.csv | cut -c1-140
