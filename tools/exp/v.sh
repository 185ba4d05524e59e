cd /root/repo
for v in 0 1 2; do DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=$v timeout 120 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 3 2>&1 | tail -1; done
DBFR_CONV2=1 DBFR_GEMM=split timeout 120 python tools/conv_bench.py --layer 0 --fam 2 --edges 650000 --reps 3 2>&1 | tail -1
DBFR_CONV2=0 DBFR_GEMM=f32 timeout 120 python tools/conv_bench.py --layer 0 --fam 2 --edges 650000 --reps 3 2>&1 | tail -1
python tools/exp/split_check.py 2>&1 | tail -2
