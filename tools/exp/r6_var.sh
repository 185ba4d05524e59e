#!/bin/bash
# developer (round 6): one variant library -- correctness on single convs + the conv tests of the GPU suite, timing against the tree, warm-unit timeline inside the bench
#   bash tools/exp/r6_var.sh <variant> [<dev variant> <waves> <cap> <sel...>]
R=$GRAFT_REPO_ROOT; cd $R
V=$1; DEVV=$2; NWV=${3:-12}; CAP=${4:-512}; shift 4
O=$R/gpurun_out/r6_var_$V.txt; : > $O
export DBFR_LIB=$R/tools/exp/ab/libdbfr_$V.so
timeout 600 python tools/exp/convz_check.py --time >> $O 2>&1
# (the suite refuses a foreign build id: test_native_library_is_loaded is left out)
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_conv_and_reduce or reduce_first or batch_independ or bitwise" 2>&1 | tail -8 >> $O
unset DBFR_LIB
bash tools/exp/bench_ab.sh 2 tree $V >> $O 2>&1
if [ -n "$DEVV" ]; then
  export DBFR_LIB=$R/tools/exp/ab/libdbfr_$DEVV.so
  Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
  for sel in "$@"; do
    echo "== bench timeline, launches whose first conv has $sel c tiles" >> $O
    rm -f $R/gpurun_out/cz_trace.bin
    DBFR_CONVZ_DEBUG_SEL=$sel DBFR_CONVZ_ABL=128 DBFR_CONVZ_DEBUG=$R/gpurun_out/cz_trace.bin timeout 300 python bench.py --steps 1 --warmup 0 $Q > /dev/null 2>&1
    python tools/exp/convz_trace.py $R/gpurun_out/cz_trace.bin $NWV $CAP 2>&1 | grep -v "first 12\|by k tile" | cut -c1-330 >> $O
  done
fi
cat $O
