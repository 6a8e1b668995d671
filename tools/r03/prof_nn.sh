# usage: bash tools/r03/prof_nn.sh TAG LIB SHAPE -- three rocprofv3 --pmc passes over tools/nn_driver.py (search kernel only matters)
set -x
TAG=$1; LIB=$2; SHAPE=${3:-c4}
OUT=gpurun_out/r03/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export GORIO_AMD_LIB=$LIB
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/a -o p -- python3 tools/nn_driver.py $SHAPE > $OUT/a.txt 2> $OUT/a.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/b -o p -- python3 tools/nn_driver.py $SHAPE > $OUT/b.txt 2> $OUT/b.err || exit 1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/c -o p -- python3 tools/nn_driver.py $SHAPE > $OUT/c.txt 2> $OUT/c.err || exit 1
python3 tools/pmc_summary.py --workload $SHAPE --csv-dir $OUT $(find $OUT -name "*counter_collection.csv" | sort) > $OUT/summary.json
grep -h "nn_search_pruned" $OUT/*_per_kernel.csv
head -1 $OUT/a_per_kernel.csv $OUT/b_per_kernel.csv $OUT/c_per_kernel.csv
