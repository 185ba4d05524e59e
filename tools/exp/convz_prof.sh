#!/bin/bash
# developer: per-kernel time and counters of the reduce-first conv pair on single convs (tools/exp/convz_check.py --timeonly)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/convz_prof; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python $R/tools/exp/convz_check.py --timeonly > $O/stats.log 2>&1
cp $O/st/*kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/st
head -5 $O/kernel_stats.csv | cut -c1-160
if [ "$1" = pmc ]; then
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o p -- python $R/tools/exp/convz_check.py --timeonly > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmc_summary.py "$O/p*/*counter_collection.csv" | head -60 > $O/pmc_summary.txt
rm -rf $O/p[0-9]
cat $O/pmc_summary.txt | head -50
fi
