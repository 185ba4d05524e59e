#!/bin/bash
# developer: one big conv (layer 3, atom-atom, 650 k edges) through every variant of the split-bf16 kernel
cd "$(dirname "$0")/../.."
run() { echo "--- $*"; env "$@" timeout 120 python tools/conv_bench.py --layer ${LAYER:-3} --fam 2 --edges ${EDGES:-650000} --reps 3 2>&1 | tail -1; }
run DBFR_CONV2=0 DBFR_GEMM=f32
run DBFR_CONV2=1 DBFR_GEMM=f32
run DBFR_CONV2=1 DBFR_GEMM=split
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=20
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=1
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=2
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=22
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=3
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2S_VAR=4
run DBFR_CONV2=1 DBFR_GEMM=split DBFR_CONV2_SKEW=4
