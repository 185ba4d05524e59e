cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_default_3.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_default_3.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d.get('value_literal_128x40'), r['fp32_equivalent_tflops'], r['frac'], r['conv_time_share'], d['latency']['cfg1_1x4']['seconds'], d['latency']['cfg2_bs16']['seconds'])
PY
