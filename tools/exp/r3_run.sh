cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
( time timeout 900 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/r3/bench_default_4.json 2> gpurun_out/r3/bench_default_4.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_default_4.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], r['fp32_equivalent_tflops'], r['frac'], r['traffic'], r['traffic_ratio'], r['traffic_counters']); print(r['traffic_source'][:200]); print(d['native_f32']['traffic'], d['native_f32']['traffic_source'][:120])
PY
tail -3 gpurun_out/r3/bench_default_4.err
