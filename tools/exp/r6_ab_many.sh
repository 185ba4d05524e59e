#!/bin/bash
# developer (round 6): single-conv correctness of each variant library, then the bench through all of them, alternating:  bash tools/exp/r6_ab_many.sh <rounds> <variant...>
R=$GRAFT_REPO_ROOT; cd $R
ROUNDS=$1; shift
for v in "$@"; do
  echo "== $v"; DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so timeout 600 python tools/exp/convz_check.py --time 2>&1 | grep -v amdgpu.ids | tail -8
done
bash tools/exp/bench_ab.sh $ROUNDS tree "$@"
