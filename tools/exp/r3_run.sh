cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
DBFR_GEMM=split_f16 DBFR_CONV2=1 timeout 120 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 10 2>&1 | tail -1 | cut -c1-200
