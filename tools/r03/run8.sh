set -x
mkdir -p gpurun_out/r03
for sp in 8 16; do
  GORIO_NN_SPLITS=$sp timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c58_sp$sp.json 2> gpurun_out/r03/c58_sp$sp.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/c58_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_8.log 2>&1
tail -5 gpurun_out/r03/gpu_tests_8.log
