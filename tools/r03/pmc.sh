# usage: bash tools/r03/pmc.sh WORKLOAD [extra bench args]   (WORKLOAD = c4 | c3 | c5)
# The rocprofv3 passes behind profiles/r03/ and profiles/kernel_counters.json: a kernel trace (+ stats) of the timed configuration and
# five --pmc passes, each in its own run with --kernel-trace only (never combined with other trace domains).  C4 is profiled in the SAME
# mode it is timed in (scan matching and GP windows overlapped).  tools/pmc_summary.py condenses the passes, tools/search_work.py adds the
# distance evaluations of the pruned search from a -DGORIO_STATS build.
set -x
WL=${1:-c4}; shift
OUT=gpurun_out/r03/pmc_$WL
mkdir -p $OUT profiles/r03
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-exhaustive $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-check --no-exhaustive $* > $OUT/trace.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS > $OUT/sq1.json 2> $OUT/sq1.err || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.json 2> $OUT/sq2.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq3 -o p -- python3 $ARGS > $OUT/sq3.json 2> $OUT/sq3.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err || exit 1
python3 tools/pmc_summary.py --workload $WL --out profiles/kernel_counters.json --csv-dir profiles/r03/$WL --skip-first 0 \
  --note "round 3; $WL collected by tools/r03/pmc.sh in the timed mode (C4: scan matching and GP windows overlapped)" \
  $(find $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/fetch $OUT/write -name "*counter_collection.csv" | sort) || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) profiles/r03/${WL}_kernel_stats.csv
cp $OUT/trace.json profiles/r03/${WL}_under_rocprof.json
