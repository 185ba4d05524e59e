#!/bin/bash
# developer (round 6): correctness of the reduce-first conv on single convs, the conv tests of the GPU suite, then timing tree vs variants
#   bash tools/exp/r6_dev.sh [variants...]   -> gpurun_out/r6_dev.txt
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r6_dev.txt; : > $O
timeout 600 python tools/exp/convz_check.py --time >> $O 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_conv_and_reduce or reduce_first or native_library or batch_independ or bitwise" 2>&1 | tail -15 >> $O
for v in "$@"; do
  export DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so
  echo "== variant $v" >> $O
  timeout 200 python tools/exp/convz_check.py --timeonly 2>&1 | grep reduce_first >> $O
done
unset DBFR_LIB
bash tools/exp/bench_ab.sh 1 tree "$@" >> $O 2>&1
cat $O
