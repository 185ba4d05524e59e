cd $GRAFT_REPO_ROOT
bash tools/exp/bench_ab.sh 3 base tree
for v in base tree; do if [ $v = tree ]; then unset DBFR_LIB; else export DBFR_LIB=$GRAFT_REPO_ROOT/tools/exp/ab/libdbfr_$v.so; fi; python bench.py --steps 3 --no-cpu-baseline --no-latency --no-native --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', d['value'], {k:r[k] for k in ('achieved','useful_tflops','padding_ratio','reference_flops_over_time_tflops','algorithmic_bytes_per_launch','fused_form_bytes_per_launch','flops_per_launch','launches')})"; done
