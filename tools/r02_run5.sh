set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_apd_gpu.py tests/test_configs_gpu.py tests/test_host_cpp.py -m gpu -x -q > gpurun_out/r02/gputest8.log 2>&1
tail -5 gpurun_out/r02/gputest8.log
python bench.py --steps 30 --warmup 3 > gpurun_out/r02/bench_e.json 2> gpurun_out/r02/bench_e.err
python bench.py --steps 10 --warmup 2 --search brute --no-cpu-baseline > gpurun_out/r02/bench_e_brute.json 2>> gpurun_out/r02/bench_e.err
