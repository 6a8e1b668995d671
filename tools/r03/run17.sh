set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_ugpm_gpu.py tests/test_golden.py tests/test_configs_gpu.py tests/test_host_cpp.py -m gpu -x -q > gpurun_out/r03/t17.log 2>&1
rc=$?
tail -3 gpurun_out/r03/t17.log
[ $rc -eq 0 ] || exit $rc
for v in base ugcap3 ugcap4; do
  L=""; [ $v != base ] && L=$PWD/tools/variants/$v.so
  GORIO_AMD_LIB=$L timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/cap_${v}_ov.json 2> gpurun_out/r03/cap_${v}_ov.err || exit 1
  GORIO_AMD_LIB=$L timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exhaustive --no-check --no-overlap > gpurun_out/r03/cap_${v}_no.json 2> gpurun_out/r03/cap_${v}_no.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/cap_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,2) for k,v in d['device_ms_per_step'].items()}, {k: round(v,4) for k,v in d['host_phase_seconds'].items()})
PY
