#!/bin/bash
# developer (round 6): the reduce-first pair side by side on disjoint compute units (DBFR_CORUN = workgroups of k_convz): poses/s by share
R=$GRAFT_REPO_ROOT; cd $R
Q="--no-cpu-baseline --no-latency --no-native --no-pmc --no-profile"
for r in 1 2; do
for n in 0 "$@"; do
  echo -n "DBFR_CORUN=$n r$r: "
  DBFR_CORUN=$n timeout 300 python bench.py --steps 3 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
done
done
