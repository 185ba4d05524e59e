#!/bin/bash
# Round-2 measurement artefacts in one GPU-box call (copied by hand into profiles/ afterwards):
#   bench lines (default, cfg 3, cfg 4, cfg 5, cfg 1), 2-rank gloo dry run of the sharded driver on one GPU,
#   rocprofv3 kernel stats of the default command, PMC passes (counters only, one pass per group) for the conv kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2prof
mkdir -p $OUT
cd $R
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in 3 4; do timeout 200 python bench.py --config $c --steps 2 --no-cpu-baseline --no-latency > $OUT/bench_cfg$c.json 2>/dev/null; done
timeout 200 python bench.py --config 5 --steps 2 --batch-poses 160 --no-cpu-baseline --no-latency > $OUT/bench_cfg5.json 2>/dev/null
timeout 300 python bench.py --config 2 --steps 2 --scaling strong --no-cpu-baseline --no-latency > $OUT/bench_strong1.json 2>/dev/null
# the N>1 path on one GPU: 2 ranks share cuda:0, records staged through the host for the gloo gather
DBFR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 1 --warmup 0 --batch-poses 320 --no-profile > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-native > $OUT/stats.log 2>&1
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats_steps2_b640.csv 2>/dev/null
rm -rf $OUT/stats
ARGS="--steps 1 --warmup 0 --batch-poses 640 --no-cpu-baseline --no-profile --no-latency --no-native"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmc_summary.py "$OUT/p*/*counter_collection.csv" | head -80 > $OUT/pmc_summary_b640.txt
python $R/tools/pmc_to_json.py $OUT/pmc_summary_b640.txt k_conv2r $OUT/pmc_k_conv2r.json
# the fp32-instruction kernels for the record (the bench line's native_f32 leg is one batch; this is the whole default command)
cd $R; DBFR_GEMM=f32 timeout 400 python bench.py --no-cpu-baseline --no-latency > $OUT/bench_gemm_f32.json 2>/dev/null; DBFR_GEMM=split_l1 timeout 400 python bench.py --steps 2 --no-cpu-baseline --no-latency --no-native > $OUT/bench_gemm_split_l1.json 2>/dev/null
rm -rf $OUT/p[0-9]
head -30 $OUT/pmc_summary_b640.txt
head -6 $OUT/kernel_stats_steps2_b640.csv
for f in $OUT/bench_*.json; do echo $f; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['scaling'], (d.get('roofline') or {}).get('achieved'), d.get('latency'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print('bad', e)
"; done
