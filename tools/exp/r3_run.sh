cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
DBFR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 8 --steps 1 --warmup 0 --batch-poses 160 --no-profile > gpurun_out/r3/bench_8rank_gloo.json 2> gpurun_out/r3/bench_8rank_gloo.err
tail -c 1500 gpurun_out/r3/bench_8rank_gloo.json | cut -c1-1500
grep -i "dbfr.dist\|error\|Traceback" gpurun_out/r3/bench_8rank_gloo.err | head -5
