cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python bench.py --steps 60 --no-cpu-baseline --no-latency --no-native --no-pmc > gpurun_out/r3/bench_soak60.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_soak60.json').read().strip().splitlines()[-1])
r=d['roofline']; print('soak 60 batches (38 400 poses):', d['value'], d['value_literal_128x40'], r['fp32_equivalent_tflops'], r['frac'])
PY
rocm-smi --showtemp 2>&1 | grep -i "junction\|hotspot\|edge" | head -3
