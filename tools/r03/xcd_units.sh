# scan pairs / clouds pinned to XCDs (GORIO_XCD_UNITS, default) against the plain block numbering: tests, then C4 and C5
set -x
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/xu_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r03/xu_tests.log
[ $rc -eq 0 ] || exit $rc
B="--steps 40 --warmup 5 --no-cpu-baseline --no-exhaustive --no-check"
for rep in 1 2; do
for v in default noxcd; do
  if [ $v = default ]; then unset GORIO_AMD_LIB; else export GORIO_AMD_LIB=$PWD/tools/variants/$v.so; fi
  timeout -k 10 300 python bench.py $B --no-overlap > gpurun_out/r03/xu_no_${v}_$rep.json 2> gpurun_out/r03/xu.err || { tail -5 gpurun_out/r03/xu.err; exit 1; }
  timeout -k 10 300 python bench.py $B > gpurun_out/r03/xu_ov_${v}_$rep.json 2> gpurun_out/r03/xu.err || exit 1
done
done
for v in default noxcd; do
  if [ $v = default ]; then unset GORIO_AMD_LIB; else export GORIO_AMD_LIB=$PWD/tools/variants/$v.so; fi
  timeout -k 10 400 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-exhaustive --no-check > gpurun_out/r03/xu_c5_${v}.json 2> gpurun_out/r03/xu.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/xu_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if not k.startswith('ugpm')})
PY
