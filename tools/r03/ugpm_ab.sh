# A/B of the rotation fit's schedule (three launches per iteration against four): the UGPM GPU tests, then the C4 step with the GP
# windows after and beside the scan matching, each schedule twice in alternation.
set -x
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_ugpm_gpu.py -m gpu -x -q > gpurun_out/r03/ugpm_tests_ab.log 2>&1
rc=$?
tail -15 gpurun_out/r03/ugpm_tests_ab.log
[ $rc -eq 0 ] || exit $rc
B="--steps 40 --warmup 5 --no-cpu-baseline --no-exhaustive --no-check"
for rep in 1 2; do
  timeout -k 10 300 python bench.py $B --no-overlap > gpurun_out/r03/uab_no3_$rep.json 2> gpurun_out/r03/uab.err || exit 1
  timeout -k 10 300 python bench.py $B --no-overlap --ugpm-four-launch > gpurun_out/r03/uab_no4_$rep.json 2> gpurun_out/r03/uab.err || exit 1
  timeout -k 10 300 python bench.py $B > gpurun_out/r03/uab_ov3_$rep.json 2> gpurun_out/r03/uab.err || exit 1
  timeout -k 10 300 python bench.py $B --ugpm-four-launch > gpurun_out/r03/uab_ov4_$rep.json 2> gpurun_out/r03/uab.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/uab_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['device_ms_per_step'].items()})
PY
