cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 600 python tools/pipeline_bench.py > gpurun_out/r3/pipeline.json 2> gpurun_out/r3/pipeline.err
tail -c 1500 gpurun_out/r3/pipeline.json; tail -3 gpurun_out/r3/pipeline.err
