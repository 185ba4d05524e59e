#!/bin/bash
# developer: bench.py (short) with an environment variable at several values, alternating: bash tools/exp/env_ab.sh <rounds> <VAR> <value...>
R=$GRAFT_REPO_ROOT; cd $R
ROUNDS=$1; VAR=$2; shift 2
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    echo -n "$VAR=$v r$r: "
    env $VAR=$v timeout 300 python bench.py --steps 3 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['frac'], r['fp32_equivalent_tflops'], r['avg_launch_ms'], r.get('conv_time_share'))"
  done
done
