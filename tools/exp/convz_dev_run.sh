#!/bin/bash
# developer: ablation timings + the s_memtime timeline of k_convz from the dev variant library (tools/exp/build_variant.py dev --dev)
R=$GRAFT_REPO_ROOT; cd $R
export DBFR_LIB=$R/tools/exp/ab/libdbfr_${1:-dev}.so
O=$R/gpurun_out/convz_dev_${1:-dev}.txt; : > $O
for a in ${2:-0 1 2 4 8 12 16 32 64}; do
  echo "== DBFR_CONVZ_ABL=$a" >> $O
  DBFR_CONVZ_ABL=$a timeout 200 python tools/exp/convz_check.py --timeonly 2>&1 | grep reduce_first >> $O
done
DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace.bin timeout 200 python tools/exp/convz_check.py --timeonly > /dev/null 2>&1
python tools/exp/convz_trace.py $R/gpurun_out/cz_trace.bin >> $O 2>&1
cat $O
