# usage: bash tools/r02_prof.sh TAG   -- kernel trace + the SQ counter pass of one non-overlapped C4 step
set -x
TAG=${1:-x}
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/trace_$TAG -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-exhaustive --no-overlap > gpurun_out/r02/trace_$TAG.json 2> gpurun_out/r02/trace_$TAG.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r02/pmc_sq_$TAG -o p -- python3 bench.py --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-check --no-exhaustive > gpurun_out/r02/pmc_sq_$TAG.json 2> gpurun_out/r02/pmc_sq_$TAG.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/r02/pmc_sq2_$TAG -o p -- python3 bench.py --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-check --no-exhaustive > gpurun_out/r02/pmc_sq2_$TAG.json 2> gpurun_out/r02/pmc_sq2_$TAG.err
