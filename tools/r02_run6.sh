set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_ugpm_gpu.py tests/test_configs_gpu.py tests/test_golden.py -m gpu -x -q > gpurun_out/r02/gputest9.log 2>&1
tail -5 gpurun_out/r02/gputest9.log
for G in 1 2 4; do
  GORIO_UGPM_GROUPS=$G python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-exhaustive > gpurun_out/r02/bench_f_g$G.json 2> gpurun_out/r02/bench_f.err
  GORIO_UGPM_GROUPS=$G python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-overlap --no-exhaustive > gpurun_out/r02/bench_f_g${G}_noov.json 2>> gpurun_out/r02/bench_f.err
done
