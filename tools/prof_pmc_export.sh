#!/bin/bash
# PMC passes for k_pose_metrics (row f3), same rules as prof_pmc.sh: each pass its own run, counters only.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_f3
ARGS="--reps 2"
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p -- python $R/tools/export_bench.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $OUT/p4 -o p -- python $R/tools/export_bench.py $ARGS > $OUT/p4.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -o p -- python $R/tools/export_bench.py $ARGS > $OUT/p1.log 2>&1
python $R/tools/pmc_summary.py "$OUT/p*/*counter_collection.csv" | head -60 > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p3 $OUT/p4
cat $OUT/summary.txt
