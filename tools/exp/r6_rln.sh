#!/bin/bash
# developer (round 6): where k_reduce_ln_layer's 0.5 ms go -- kernel stats of a short bench through timing-only variants (wrong results): no LayerNorm arithmetic / one row per node
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6rln; mkdir -p $OUT
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for v in tree "$@"; do
  if [ $v = tree ]; then unset DBFR_LIB; else export DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v} -o s -- python $R/bench.py --steps 1 --warmup 0 $Q > $OUT/${v}.log 2>&1
  echo "== $v"; grep "k_reduce_ln_layer\|k_mlp\|k_conv2h\|k_convz<" $OUT/$v/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
  rm -rf $OUT/$v
done
