set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_apd_gpu.py tests/test_golden.py tests/test_real_clouds.py tests/test_configs_gpu.py tests/test_host_cpp.py -m gpu -x -q > gpurun_out/r02/gputest4.log 2>&1
tail -12 gpurun_out/r02/gputest4.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace_knn2 -o t -- python3 tools/knn_variants.py > gpurun_out/r02/trace_knn2.log 2>&1
grep -i "knn\|linearize\|nn_search" gpurun_out/r02/trace_knn2/t_kernel_stats.csv | cut -c1-150
python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_c.json 2> gpurun_out/r02/bench_c.err
