set -x
mkdir -p gpurun_out/r03
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
for v in nn_s8w5b64 nn_s8w4b64 nn_s12w4b64; do
  GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab10_${v}_plan.json 2> gpurun_out/r03/ab10_${v}_plan.err || exit 1
  GORIO_NN_NO_PLAN=1 GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab10_${v}_noplan.json 2> gpurun_out/r03/ab10_${v}_noplan.err || exit 1
done
for v in nn_s8w4b64 nn_s12w4b64; do
  GORIO_AMD_LIB=$PWD/tools/variants/$v.so timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c510_${v}_plan.json 2> gpurun_out/r03/c510_${v}_plan.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab10_*.json')+glob.glob('gpurun_out/r03/c510_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
    except Exception as e: print(f, 'ERR', e)
PY
