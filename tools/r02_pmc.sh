# usage: bash tools/r02_pmc.sh TAG -- the five rocprofv3 counter passes (each in its own run, never with the trace domains) and a
# kernel trace of one non-overlapped C4 step, for tools/pmc_summary.py
set -x
TAG=${1:-final}
OUT=gpurun_out/r02/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 1 --warmup 1 --no-overlap --no-cpu-baseline --no-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 5 --warmup 1 --no-overlap --no-cpu-baseline --no-check > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o t -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-check --no-exhaustive > $OUT/trace_overlap.json 2> $OUT/trace_overlap.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS > $OUT/sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.json 2> $OUT/sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq3 -o p -- python3 $ARGS > $OUT/sq3.json 2> $OUT/sq3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
ls $OUT/*/
