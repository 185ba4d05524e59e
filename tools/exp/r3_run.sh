mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT
for m in split_f16 split f32; do
  (DBFR_GEMM=$m DBFR_CONV2=1 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 5000 > gpurun_out/r3/power_$m.txt 2>&1 &)
  sleep 11
  echo "mode $m"
  for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 1; done
  sleep 25
  tail -1 gpurun_out/r3/power_$m.txt | cut -c1-90
done
