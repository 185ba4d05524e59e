#!/bin/bash
# developer (round 6): warm-unit timeline of a dev variant library inside the bench:  bash tools/exp/r6_tl.sh <dev variant> <waves> <cap> <sel...>
R=$GRAFT_REPO_ROOT; cd $R
DEVV=$1; NWV=$2; CAP=$3; shift 3
export DBFR_LIB=$R/tools/exp/ab/libdbfr_$DEVV.so
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for sel in "$@"; do
  echo "== bench timeline, launches whose first conv has $sel c tiles"
  rm -f $R/gpurun_out/cz_trace.bin
  DBFR_CONVZ_DEBUG_SEL=$sel DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace.bin timeout 300 python bench.py --steps 1 --warmup 0 $Q > /dev/null 2>&1
  python tools/exp/convz_trace.py $R/gpurun_out/cz_trace.bin $NWV $CAP 2>&1 | grep -v "first 12\|by k tile" | cut -c1-330
done
