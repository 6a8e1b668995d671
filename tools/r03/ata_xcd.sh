# J^T J launches: all tile groups of a window on one XCD (default) against the round-robin placement (-DGORIO_ATA_NO_XCD)
# needs the comparison build first:  make -C go-rio_amd/csrc OUT=../../tools/variants/ata_noxcd.so EXTRA=-DGORIO_ATA_NO_XCD
set -x
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_ugpm_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r03/ata_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r03/ata_tests.log
[ $rc -eq 0 ] || exit $rc
B="--steps 40 --warmup 5 --no-cpu-baseline --no-exhaustive --no-check"
for rep in 1 2; do
for v in default ata_noxcd; do
  if [ $v = default ]; then unset GORIO_AMD_LIB; else export GORIO_AMD_LIB=$PWD/tools/variants/$v.so; fi
  timeout -k 10 300 python bench.py $B --no-overlap > gpurun_out/r03/ax_no_${v}_$rep.json 2> gpurun_out/r03/ax.err || { tail -5 gpurun_out/r03/ax.err; exit 1; }
  timeout -k 10 300 python bench.py $B > gpurun_out/r03/ax_ov_${v}_$rep.json 2> gpurun_out/r03/ax.err || exit 1
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/ax_*.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],3), round(d['value']), {k:round(v,3) for k,v in d['device_ms_per_step'].items() if k.startswith('ugpm')})
PY
