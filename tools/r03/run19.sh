set -x
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_19.log 2>&1
rc=$?
tail -3 gpurun_out/r03/gpu_tests_19.log
[ $rc -eq 0 ] || exit $rc
B="--steps 20 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check"
timeout -k 10 300 python bench.py $B --no-overlap > gpurun_out/r03/ab19_no.json 2> gpurun_out/r03/ab19_no.err || exit 1
timeout -k 10 300 python bench.py $B > gpurun_out/r03/ab19_ov.json 2> gpurun_out/r03/ab19_ov.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ab19_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['device_ms_per_step'].items()}, {k: round(v,4) for k,v in d['host_phase_seconds'].items()})
PY
