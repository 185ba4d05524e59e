#!/bin/bash
# developer (round 6): rocprofv3 kernel stats of a short bench run through the tree's library and through variant libraries:  bash tools/exp/r6_stats.sh [variants...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6stats; mkdir -p $OUT
Q="--no-cpu-baseline --no-latency --no-native --no-pmc"
for v in tree "$@"; do
  if [ $v = tree ]; then unset DBFR_LIB; else export DBFR_LIB=$R/tools/exp/ab/libdbfr_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_stats -o s -- python $R/bench.py --steps 2 $Q > $OUT/${v}_stats.log 2>&1
  cp $OUT/${v}_stats/*kernel_stats.csv $OUT/${v}_kernel_stats.csv 2>/dev/null
  # per-launch durations of k_convz in launch order (the seven launches of a denoise step differ: layer depth)
  python - <<PY > $OUT/${v}_convz_by_launch.txt
import csv, glob, collections
f = glob.glob("$OUT/${v}_stats/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
for k in ("k_convz", "k_conv2h"):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if k + "<" in r["Kernel_Name"]]
    d.sort()
    per = 7 if k == "k_convz" else 6
    n = len(d) // per * per
    by = collections.defaultdict(list)
    for i, (_, dur) in enumerate(d[:n]): by[i % per].append(dur)
    print(k, "launches", len(d), "mean us by position in the step:", [round(sum(v) / len(v) / 1e3, 1) for _, v in sorted(by.items())])
PY
  rm -rf $OUT/${v}_stats
  echo "== $v"; head -4 $OUT/${v}_kernel_stats.csv | cut -c1-150; cat $OUT/${v}_convz_by_launch.txt
done

# the reduce-first kernel's ablations inside the bench (developer library): conv pair per launch
export DBFR_LIB=$R/tools/exp/ab/libdbfr_dev.so
if [ -f $DBFR_LIB ]; then
  cd $R
  for a in 0 12 32; do
    echo -n "dev ABL=$a: "
    DBFR_CONVZ_ABL=$a timeout 300 python bench.py --steps 2 $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_ms'])"
  done
fi
