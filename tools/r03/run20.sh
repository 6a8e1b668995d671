set -x
mkdir -p gpurun_out/r03
B="--steps 10 --warmup 2 --no-overlap --no-cpu-baseline --no-exhaustive --no-check"
i=0
for cfg in "2 10240 1" "4 20480 1" "4 40960 1" "3 15360 1" "2 10240 2" "4 20480 2" "2 5120 1"; do
  set -- $cfg
  GORIO_PLAN_CAPMUL=$1 GORIO_PLAN_DIV=$2 GORIO_FIRST_SPLITS=$3 timeout -k 10 300 python bench.py $B > gpurun_out/r03/plan_$1_$2_$3.json 2> gpurun_out/r03/plan_$1_$2_$3.err || exit 1
done
GORIO_PLAN_CAPMUL=4 GORIO_PLAN_DIV=20480 timeout -k 10 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/planc5_4_20480.json 2> gpurun_out/r03/planc5.err || exit 1
GORIO_PLAN_CAPMUL=2 GORIO_PLAN_DIV=10240 timeout -k 10 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/planc5_2_10240.json 2> gpurun_out/r03/planc5.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/plan_*.json')+glob.glob('gpurun_out/r03/planc5_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
PY
