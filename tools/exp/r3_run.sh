cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 1 > gpurun_out/r3/bench_final.json 2> gpurun_out/r3/bench_final.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['value_literal_128x40'], d['ms_per_step'], r['achieved'], r['frac'], r['fp32_equivalent_tflops'], r['traffic'], r['traffic_ratio'], d['cpu_baseline']['value'], d['cpu_baseline']['parity']['lig_rmsd_A'], d['native_f32']['poses_per_sec'])
PY
