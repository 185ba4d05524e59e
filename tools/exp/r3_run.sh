cd $GRAFT_REPO_ROOT
for c in cfg1 bs16 bs40 p80 p160 c5p16; do for ns in 1 0; do echo -n "$c nosplit-old-rule=$ns: "; DBFR_CONV2_NOSPLIT=$ns timeout 200 python tools/latency_run.py --case $c --reps 5 2>&1 | tail -1; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_kernels or bitwise or fused" 2>&1 | tail -2
