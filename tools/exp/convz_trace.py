"""Developer: read the s_memtime timeline k_convz<8,128> wrote (DBFR_CONVZ_DEBUG=<file> DBFR_CONVZ_ABL=128, developer build) and print, per wave,
the cycles between the stamps of the tile loop: 10 tile start, 11 step A done, 12 step B done, 13 barrier passed."""
import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8, 1024)
for w in range(8):
    v = t[w][t[w] > 0]
    tags, ts = (v & 0xff).astype(int), (v >> 8).astype(np.int64)
    if len(ts) == 0:
        continue
    print(f"wave {w}: {len(ts)} stamps; prologue: slots {ts[1]-ts[0] if len(ts)>1 else 0}  hidden+scan {ts[2]-ts[1] if len(ts)>2 else 0}")
    # tile loop
    i = 3; rows = []
    while i + 2 < len(ts) and len(rows) < 400:
        if tags[i] == 10:
            seq = {}
            k = 1
            while i + k < len(ts) and tags[i + k] != 10:
                seq[int(tags[i + k])] = int(ts[i + k]); k += 1
            rows.append((seq.get(11, ts[i]) - ts[i], seq.get(12, ts[i]) - ts[i], seq.get(13, ts[i]) - ts[i]))
            i += k
        else:
            i += 1
    r = np.array(rows)
    if len(r):
        first = "A first" if w < 4 else "B first"
        print(f"   {first}: tiles {len(r)}; mean cycles to [A done, B done, barrier passed] = {r.mean(0).round(0).tolist()}   median {np.median(r,0).tolist()}")
        print("   first 12 tiles:", r[:12].tolist())
