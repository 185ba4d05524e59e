#!/bin/bash
# developer: bench.py (short) through variant libraries, alternating: bash tools/exp/bench_ab.sh <rounds> <variant...>   ("tree" = the tree's library)
R=$GRAFT_REPO_ROOT; cd $R
ROUNDS=$1; shift
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ $v = tree ]; then unset DBFR_LIB; else export DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so; fi
    echo -n "$v r$r: "
    timeout 300 python bench.py --steps 3 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['fp32_equivalent_tflops'], r['avg_launch_ms'], r.get('conv_time_share'))"
  done
done
