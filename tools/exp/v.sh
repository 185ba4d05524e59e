cd /root/repo
for a in 0 1 2 4; do echo "ABL $a"; DBFR_CONV2R_ABL=$a DBFR_CONV2_RING=1 DBFR_CONV2=1 DBFR_GEMM=split timeout 120 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 3 2>&1 | tail -1; done
DBFR_CONV2_RING=1 DBFR_CONV2=1 DBFR_GEMM=split timeout 120 python tools/conv_bench.py --layer 0 --fam 2 --edges 650000 --reps 3 2>&1 | tail -1
DBFR_CONV2_RING=1 DBFR_CONV2=1 DBFR_GEMM=split timeout 120 python tools/conv_bench.py --layer 3 --fam 2 --edges 777 --reps 3 2>&1 | tail -1
