cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/lat -o s -- python $R/tools/latency_run.py --case cfg1 --reps 5 > $R/gpurun_out/r3/lat.log 2>&1
tail -2 $R/gpurun_out/r3/lat.log | head -1
head -14 $R/gpurun_out/r3/lat/*kernel_stats.csv | cut -c1-130
python - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(glob.glob(R+'/gpurun_out/r3/lat/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last run: take last 1/6 of the kernels
n=len(rows)//6
last=rows[-n:]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in last)
span=int(last[-1]['End_Timestamp'])-int(last[0]['Start_Timestamp'])
print('last run: kernels',n,'busy ms',busy/1e6,'span ms',span/1e6)
PY
rm -rf $R/gpurun_out/r3/lat
