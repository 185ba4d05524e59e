mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
SKIP_B=1 timeout 200 tools/exp/split_f16 > gpurun_out/r3/split_f16_v2.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/test_full_1.txt
cat gpurun_out/r3/test_full_1.txt
timeout 900 python bench.py > gpurun_out/r3/bench_default_1.json 2> gpurun_out/r3/bench_default_1.err
tail -c 3000 gpurun_out/r3/bench_default_1.json
tail -5 gpurun_out/r3/bench_default_1.err
