"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches)."""
import csv
import glob
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in sys.argv[1:]:
    for f in glob.glob(path, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:48]
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", out[k].get("FETCH_SIZE", 0))):
    print(f"{k}  dispatches={len(calls[k])}")
    for c, v in sorted(out[k].items()):
        print(f"    {c:32s} {v:.6g}")
