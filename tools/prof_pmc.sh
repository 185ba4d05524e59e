#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun). Each pass is its own run,
# counters only (never combined with trace domains other than the implicit kernel dispatch record).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
ARGS="--steps 1 --warmup 0 --batch-poses ${2:-160} --no-cpu-baseline --no-profile"
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -o p -- python $R/bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p2 -o p -- python $R/bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p -- python $R/bench.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT/p4 -o p -- python $R/bench.py $ARGS > $OUT/p4.log 2>&1
python $R/tools/pmc_summary.py "$OUT/p*/*counter_collection.csv" | head -60 > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/summary.txt
