set -x
mkdir -p gpurun_out/r02 && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 2 > gpurun_out/r02/bench_a.json 2> gpurun_out/r02/bench_a.err
tail -c 600 gpurun_out/r02/bench_a.err
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r02/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace_a -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-exhaustive > gpurun_out/r02/trace_a.json 2> gpurun_out/r02/trace_a.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/pmc_sq_a -o p -- python3 bench.py --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-check > gpurun_out/r02/pmc_sq_a.json 2> gpurun_out/r02/pmc_sq_a.err
ls gpurun_out/r02/pmc_sq_a | head
