set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_apd_gpu.py tests/test_golden.py tests/test_real_clouds.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r02/gputest3.log 2>&1
tail -15 gpurun_out/r02/gputest3.log
python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-overlap --no-exhaustive > gpurun_out/r02/bench_b_noov.json 2>> gpurun_out/r02/bench_b.err
tail -c 300 gpurun_out/r02/bench_b.err
