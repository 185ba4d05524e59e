#!/bin/bash
# developer: the warm-unit timeline of ROUND 5's k_convz (variant r5t) inside the bench, for comparison with tools/exp/r6_trace.sh
R=$GRAFT_REPO_ROOT; cd $R
export DBFR_LIB=$R/tools/exp/ab/libdbfr_r5t.so
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
O=$R/gpurun_out/r6_trace_r5.txt; : > $O
for sel in "$@"; do
  echo "== bench, launches whose first conv has $sel c tiles" >> $O
  rm -f $R/gpurun_out/cz_trace_r5.bin
  DBFR_CONVZ_DEBUG_SEL=$sel DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace_r5.bin timeout 300 python bench.py --steps 1 --warmup 0 $Q > /dev/null 2>&1
  python tools/exp/convz_trace.py $R/gpurun_out/cz_trace_r5.bin 8 1024 2>&1 | cut -c1-500 >> $O
done
cat $O
