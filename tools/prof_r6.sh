#!/bin/bash
# Round-6 measurement artefacts in one GPU-box call (copied by hand into profiles/ afterwards):
#   rocprofv3 kernel stats of the default command, PMC passes (counters only, one pass per group) for the conv pair k_convz + k_conv2h,
#   cfg 5 at its stated size (64 jobs x 40 poses): bench line, kernel stats, PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6prof
mkdir -p $OUT
WHAT=${1:-all}
pmc() {   # pmc <tag> <bench args...>
  local tag=$1; shift
  local i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
             "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $grp --output-format csv -d $OUT/${tag}_p$i -o p -- python $R/bench.py "$@" > $OUT/${tag}_p$i.log 2>&1 || echo "$tag pass $i failed"
  done
  python $R/tools/pmc_summary.py "$OUT/${tag}_p*/*counter_collection.csv" | head -120 > $OUT/${tag}_pmc_summary.txt
  python $R/tools/pmc_to_json.py $OUT/${tag}_pmc_summary.txt k_conv2h $OUT/${tag}_pmc_k_conv2h.json
  python $R/tools/pmc_to_json.py $OUT/${tag}_pmc_summary.txt k_convz $OUT/${tag}_pmc_k_convz.json
  rm -rf $OUT/${tag}_p[0-9]
}
stats() {  # stats <tag> <bench args...>
  local tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -o s -- python $R/bench.py "$@" > $OUT/${tag}_stats.log 2>&1
  cp $OUT/${tag}_stats/*kernel_stats.csv $OUT/${tag}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/${tag}_stats
  head -8 $OUT/${tag}_kernel_stats.csv
}
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
if [ $WHAT = all ] || [ $WHAT = lines ]; then
  cd $R
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  for c in 3 4; do timeout 300 python bench.py --config $c --steps 2 $Q > $OUT/bench_cfg$c.json 2>/dev/null; done
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 1 > $OUT/bench_driver_style_steps20.json 2> $OUT/bench_driver_style_steps20.err
  timeout 300 python bench.py --config 2 --steps 2 --scaling strong $Q > $OUT/bench_strong1.json 2>/dev/null
  timeout 300 python bench.py --config 1 --batch-poses 4 --steps 1 --no-native --no-latency > $OUT/bench_cfg1_1x4.json 2>/dev/null
  # the N>1 path on one GPU: 2 ranks share cuda:0, records staged through the host for the gloo gather
  DBFR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 1 --warmup 0 --batch-poses 320 --no-profile > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
  # the same WITHOUT a launcher: bench.py starts its own two ranks
  DBFR_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 1 --warmup 0 --batch-poses 320 --no-profile > $OUT/bench_2rank_selfspawn_gloo.json 2> $OUT/bench_2rank_selfspawn_gloo.err
  for m in split_f16 f32; do DBFR_GEMM=$m timeout 400 python bench.py --steps 2 $Q > $OUT/bench_gemm_$m.json 2>/dev/null; done
  for f in $OUT/bench_*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print(d['value'], d['n_gpus'], d['scaling'], r.get('fp32_equivalent_tflops'), r.get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('parity'))
except Exception as e: print('bad', e)
"; done
  cd /tmp
fi
if [ $WHAT = all ] || [ $WHAT = fullsize ]; then
  # one rank's share of the sharded configs at their STATED size, through the nccl backend (one rank), records streamed to pinned host memory,
  # gather to rank 0: cfg 4 = all 2 000 pockets x 40 poses (80 k poses); cfg 3 = 1 250 of the 10 000 ligands (one of eight shares) x 40 poses
  cd $R
  DBFR_DIST_SINGLE=1 timeout 900 python bench.py --config 4 --jobs 2000 --store host --gather root $Q > $OUT/bench_cfg4_full.json 2> $OUT/bench_cfg4_full.err
  DBFR_DIST_SINGLE=1 timeout 900 python bench.py --config 3 --jobs 1250 --store host --gather root $Q > $OUT/bench_cfg3_share.json 2> $OUT/bench_cfg3_share.err
  for f in $OUT/bench_cfg4_full.json $OUT/bench_cfg3_share.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['steps'], d['config']['per_rank'])"; done
  cd /tmp
fi
if [ $WHAT = all ] || [ $WHAT = cfg2 ]; then
  stats cfg2 --steps 2 --warmup 1 $Q
  pmc cfg2 --steps 1 --warmup 0 --batch-poses 640 --no-profile $Q
fi
if [ $WHAT = all ] || [ $WHAT = cfg5 ]; then
  cd $R; timeout 900 python bench.py --config 5 --steps 4 --warmup 1 --batch-poses 640 --no-latency --no-native --cpu-batched-steps 0 > $OUT/bench_cfg5_64x40.json 2> $OUT/bench_cfg5_64x40.err; cd /tmp
  tail -c 600 $OUT/bench_cfg5_64x40.err
  stats cfg5 --config 5 --steps 1 --warmup 1 --batch-poses 640 $Q
  pmc cfg5 --config 5 --steps 1 --warmup 0 --batch-poses 640 --no-profile $Q
fi
if [ $WHAT = all ] || [ $WHAT = examples ]; then
  # the reference's examples graph by graph, with the library's own near-tie read-out
  cd $R; timeout 900 python -m pytest tests/test_examples.py -q -m gpu -s 2>&1 | grep -v "^\s*$" > $OUT/examples_graph_by_graph.txt; tail -3 $OUT/examples_graph_by_graph.txt; cd /tmp
fi
if [ $WHAT = all ] || [ $WHAT = bs ]; then
  # poses/s against the batch size predict.py hands the model (-bs): what INTEGRATION.md tells its users
  cd $R
  for c in bs16 bs32 bs64 bs128 bs256 p160 p320 p640; do timeout 200 python tools/latency_run.py --case $c 2>/dev/null | tail -1; done > $OUT/poses_per_sec_by_batch_size.txt 2>&1
  cat $OUT/poses_per_sec_by_batch_size.txt; cd /tmp
fi
ls $OUT
