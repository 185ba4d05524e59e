#!/bin/bash
# developer (round 6): the reduce-first pair one after the other vs k_conv2h on a side stream (DBFR_PAIR_OVERLAP=1): poses (sha256) and seconds per call by batch size, the bench
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for o in 0 1; do echo "== DBFR_PAIR_OVERLAP=$o"; DBFR_PAIR_OVERLAP=$o python tools/exp/pose_hash.py cfg1 bs16 bs64 p160 cfg1x40 c5p16 2>&1 | grep -v amdgpu.ids; done
done
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for r in 1 2; do for o in 0 1; do echo -n "bench overlap=$o: "; DBFR_PAIR_OVERLAP=$o timeout 300 python bench.py --steps 3 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done; done
