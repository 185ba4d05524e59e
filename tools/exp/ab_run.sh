#!/bin/bash
# A/B of two builds of the library in ONE GPU-box call (same board, same session): tools/exp/ab/libdbfr_base.so vs the tree's libdbfr.so.
#   bash tools/exp/ab_run.sh <tag>      -> gpurun_out/ab_<tag>.txt
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab_${1:-x}.txt; : > $OUT
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for round in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export DBFR_LIB=$R/tools/exp/ab/libdbfr_base.so; else unset DBFR_LIB; fi
    for spec in "3 2 650000" "0 0 650000" "1 1 650000"; do set -- $spec
      echo -n "$which r$round " >> $OUT
      DBFR_CONV2=1 DBFR_GEMM=split_f16 timeout 120 python tools/conv_bench.py --layer $1 --fam $2 --edges $3 --reps 30 2>&1 | tail -1 | cut -c1-60,250-400 >> $OUT
    done
    echo -n "$which r$round bench " >> $OUT
    timeout 300 python bench.py --steps 4 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['fp32_equivalent_tflops'], r['avg_launch_ms'], r['conv_time_share'])" >> $OUT
  done
done
cat $OUT
