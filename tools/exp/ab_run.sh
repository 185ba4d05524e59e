#!/bin/bash
# A/B of several builds of the library in ONE GPU-box call (same board, same session): every tools/exp/ab/libdbfr_*.so and the tree's libdbfr.so ("tree").
#   bash tools/exp/ab_run.sh <tag> [rounds]     -> gpurun_out/ab_<tag>.txt
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/ab_${1:-x}.txt; : > $OUT
ROUNDS=${2:-2}
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
LIBS="tree $(ls tools/exp/ab/libdbfr_*.so 2>/dev/null)"
for round in $(seq 1 $ROUNDS); do
  for lib in $LIBS; do
    if [ $lib = tree ]; then unset DBFR_LIB; name=tree; else export DBFR_LIB=$R/$lib; name=$(basename $lib .so | sed s/libdbfr_//); fi
    for spec in "3 2 650000" "0 0 650000"; do set -- $spec
      echo -n "$name r$round " >> $OUT
      DBFR_CONV2=1 DBFR_GEMM=split_f16 timeout 120 python tools/conv_bench.py --layer $1 --fam $2 --edges $3 --reps 30 2>&1 | tail -1 | sed -E 's/\(all [^)]*\)//' >> $OUT
    done
    echo -n "$name r$round bench " >> $OUT
    timeout 300 python bench.py --steps 4 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['fp32_equivalent_tflops'], r['avg_launch_ms'], r['conv_time_share'])" >> $OUT
  done
done
cat $OUT
