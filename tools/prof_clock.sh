#!/bin/bash
# effective shader clock under the k_conv load: GRBM_GUI_ACTIVE / dispatch duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clock
mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o p -- python $R/bench.py --steps 1 --warmup 0 --batch-poses 320 --no-cpu-baseline --no-profile > $OUT/log 2>&1
ls $OUT/p
head -2 $OUT/p/*counter_collection.csv | cut -c1-400
python - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
cc=list(csv.DictReader(open(glob.glob(R+"/gpurun_out/clock/p/*counter_collection.csv")[0])))
kt={r["Dispatch_Id"]:r for r in csv.DictReader(open(glob.glob(R+"/gpurun_out/clock/p/*kernel_trace.csv")[0]))}
tot_c=tot_ns=0
for r in cc:
    if not r["Kernel_Name"].startswith("void k_conv<144"): continue
    k=kt.get(r["Dispatch_Id"])
    if not k: continue
    ns=int(k["End_Timestamp"])-int(k["Start_Timestamp"])
    tot_c+=float(r["Counter_Value"]); tot_ns+=ns
print("k_conv<144>: GRBM_GUI_ACTIVE sum", tot_c, "duration ns", tot_ns, "=> effective clock GHz", tot_c/tot_ns)
PY
rm -rf $OUT/p
