set -x
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_21.log 2>&1
rc=$?
tail -3 gpurun_out/r03/gpu_tests_21.log
[ $rc -eq 0 ] || exit $rc
B="--steps 20 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B --no-overlap > gpurun_out/r03/ab21_no.json 2> gpurun_out/r03/ab21_no.err || exit 1
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab21_ov.json 2> gpurun_out/r03/ab21_ov.err || exit 1
timeout -k 10 300 python bench.py --latency --latency-reps 50 > gpurun_out/r03/lat21.json 2> gpurun_out/r03/lat21.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab21_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')}, {k: round(v/20*1e3,3) for k,v in d['host_phase_seconds'].items()})
for e in json.load(open('gpurun_out/r03/lat21.json'))['latency']: print(e['shape'], round(e['median_ms'],3), round(e['set_inputs_ms'],3), round(e['align_ms'],3))
PY
