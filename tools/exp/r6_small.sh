#!/bin/bash
# developer (round 6): where does a SMALL batch spend its time?  kernel trace of predict.py-sized batches, busy share of the GPU, gaps between
# kernels, kernel time by name:   bash tools/exp/r6_small.sh [cases...]      (is it launches or kernel time?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6small; mkdir -p $OUT
for c in ${@:-bs16 cfg1x40 cfg1 bs64}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o s -- python $R/tools/latency_run.py --case $c --reps 3 > $OUT/$c.log 2>&1
  grep "poses/s" $OUT/$c.log
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$c/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the timed region = the last 3 of 4 runs: find the run boundaries by k_time_embed-free heuristic: split the kernel list in 4 equal parts by count
n = len(iv) // 4
part = iv[n:]            # runs 2..4
span = part[-1][1] - part[0][0]
busy = 0; cur_s, cur_e = part[0][0], part[0][1]
gaps = []
for s, e, _ in part[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in part:
    k = k.split("(")[0].replace("void ", "")
    by[k][0] += e - s; by[k][1] += 1
tot = sum(v[0] for v in by.values())
gaps.sort()
print(f"$c: kernels in 3 runs {len(part)}, span {span/3e6:.2f} ms per run, GPU busy {busy/3e6:.2f} ms per run ({100*busy/span:.1f} %), "
      f"gaps: n={len(gaps)} median {gaps[len(gaps)//2]/1e3:.1f} us, sum {sum(gaps)/3e6:.2f} ms per run, "
      f"largest {[round(g/1e3) for g in gaps[-5:]]} us")
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"   {k:40s} {100*t/tot:5.1f} %  {t/c/1e3:8.1f} us x {c//3} per run")
PY
  rm -rf $OUT/$c
done
