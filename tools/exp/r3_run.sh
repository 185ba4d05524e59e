mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for abl in 0 256; do DBFR_CONV2H_ABL=$abl DBFR_GEMM=split_f16 DBFR_CONV2=1 timeout 120 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 10 2>&1 | tail -1 | sed "s/^/abl $abl: /"; done; done > gpurun_out/r3/conv_bench_7.txt 2>&1
DBFR_GEMM=split_f16 DBFR_CONV2=1 timeout 120 python tools/conv_bench.py --layer 0 --fam 2 --edges 650000 --reps 10 2>&1 | tail -1 >> gpurun_out/r3/conv_bench_7.txt
cat gpurun_out/r3/conv_bench_7.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split or gemm or fixture or fused" 2>&1 | tail -5 > gpurun_out/r3/test_8.txt
cat gpurun_out/r3/test_8.txt
