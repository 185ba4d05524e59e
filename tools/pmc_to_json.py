"""profiles/r2_pmc_<kernel>.json from the per-kernel PMC summary tools/pmc_summary.py prints (tools/prof_r2.sh):
    python tools/pmc_to_json.py gpurun_out/r2prof/pmc_summary_b640.txt k_conv2r profiles/r2_pmc_k_conv2r.json
FETCH_SIZE / WRITE_SIZE are KiB summed over the kernel's dispatches; traffic = raw fetch + raw write per launch."""
import json
import re
import sys

txt, kernel, out = sys.argv[1:4]
cur, vals, disp = None, {}, 0
for line in open(txt):
    m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
    if m:
        cur = m.group(1)
        if kernel in cur:
            disp = int(m.group(2))
        continue
    if cur and kernel in cur and line.startswith("    "):
        k, v = line.split()
        vals[k] = float(v)
assert disp and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals, (disp, sorted(vals))
fetch, write = vals["FETCH_SIZE"] * 1024 / disp, vals["WRITE_SIZE"] * 1024 / disp
d = {"kernel": kernel, "dispatches": disp,
     "command": "tools/prof_r2.sh: rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE TCC_HIT TCC_MISS | ... (separate passes, counters only) -- "
                "python bench.py --steps 1 --warmup 0 --batch-poses 640 --no-cpu-baseline --no-profile --no-latency --no-native",
     "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch_raw": write, "traffic_bytes_per_launch": fetch + write,
     "note": "FETCH_SIZE/WRITE_SIZE in KiB summed over dispatches; gfx950 FETCH_SIZE under-reports wide coalesced streams by up to 2x "
             "(MI355X_MICROARCH.md HBM section), so the corrected read side lies between 1x and 2x of raw. One launch = all four convs of an "
             "interaction layer (or the two torsion-head convs): the figure averages over both kinds. Valid for `bench.py --config 2 "
             "--batch-poses 640` only; bench.py reports it as a constant from this file, not as a measurement of its own run.",
     "counters": vals}
json.dump(d, open(out, "w"), indent=1)
print(out, f"fetch {fetch / 1e9:.2f} GB + write {write / 1e9:.2f} GB per launch over {disp} launches")
