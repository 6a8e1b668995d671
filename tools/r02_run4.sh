set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_ugpm_gpu.py tests/test_golden.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r02/gputest5.log 2>&1
tail -5 gpurun_out/r02/gputest5.log
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r02/bench_d.json 2> gpurun_out/r02/bench_d.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-overlap --no-exhaustive > gpurun_out/r02/bench_d_noov.json 2>> gpurun_out/r02/bench_d.err
