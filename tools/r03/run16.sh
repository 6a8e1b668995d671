set -x
mkdir -p gpurun_out/r03
for pr in 0 1 2; do
  GORIO_PRIO=$pr timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/prio_$pr.json 2> gpurun_out/r03/prio_$pr.err || exit 1
done
timeout -k 10 300 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/c3_plancap.json 2> gpurun_out/r03/c3_plancap.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/prio_*.json'))+['gpurun_out/r03/c3_plancap.json']:
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,2) for k,v in d['device_ms_per_step'].items()})
PY
