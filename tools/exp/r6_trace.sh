#!/bin/bash
# developer (round 6): s_memtime timeline of k_convz (developer library) in the single-conv timing run and inside the bench, per launch type (c tiles of the launch's first conv)
R=$GRAFT_REPO_ROOT; cd $R
export DBFR_LIB=$R/tools/exp/ab/libdbfr_dev.so
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
O=$R/gpurun_out/r6_trace.txt; : > $O
echo "== single conv (convz_check --timeonly)" >> $O
DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace.bin timeout 200 python tools/exp/convz_check.py --timeonly > /dev/null 2>&1
python tools/exp/convz_trace.py $R/gpurun_out/cz_trace.bin 2>&1 | cut -c1-700 >> $O
for sel in "$@"; do
  echo "== bench, launches whose first conv has $sel c tiles" >> $O
  rm -f $R/gpurun_out/cz_trace.bin
  DBFR_CONVZ_DEBUG_SEL=$sel DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace.bin timeout 300 python bench.py --steps 1 --warmup 0 $Q > /dev/null 2>&1
  python tools/exp/convz_trace.py $R/gpurun_out/cz_trace.bin 2>&1 | cut -c1-700 >> $O
done
cat $O
