set -x
mkdir -p gpurun_out/r03
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab11_plan.json 2> gpurun_out/r03/ab11_plan.err || exit 1
GORIO_NN_NO_PLAN=1 timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab11_noplan.json 2> gpurun_out/r03/ab11_noplan.err || exit 1
GORIO_NN_NO_PLAN=1 GORIO_AMD_LIB=$PWD/tools/variants/nn_nowork.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab11_nowork.json 2> gpurun_out/r03/ab11_nowork.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab11_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
