#!/bin/bash
# PMC passes over the conv micro-benchmark (tools/conv_bench.py), counters only, one pass per counter group.
#   tools/pmc_conv.sh <tag> [conv_bench args...]      (environment knobs DBFR_CONV2 ... are inherited)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmcconv_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $R/tools/conv_bench.py --reps 2 "$@" > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python $R/tools/pmc_summary.py "$OUT/p*/*counter_collection.csv" > $OUT/summary.txt
rm -rf $OUT/p[0-9]
grep -A40 "k_conv" $OUT/summary.txt | head -45
