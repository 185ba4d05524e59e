"""Developer: read the s_memtime timeline k_convz<128> wrote (DBFR_CONVZ_DEBUG=<file> DBFR_CONVZ_ABL=128, developer build) and print, per wave,
the cycles between the stamps of workgroup 0's first unit: 1 unit start, 2 slots done, 3 hidden layer done, 10 tile start, 11 step A done,
12 step B done, 13 barrier passed."""
import sys, numpy as np
NWV, CAP = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 512)      # [wave][stamp]: role-split kernel 12 x 512 (waves 0..7 chunk waves, 8..11 column waves); round 5's kernel 8 x 1024
t = np.fromfile(sys.argv[1], dtype=np.uint64)[:NWV * CAP].reshape(NWV, CAP)
for w in range(NWV):
    v = t[w][t[w] > 0]
    tags, ts = (v & 0xff).astype(int), (v >> 8).astype(np.int64)
    if len(ts) == 0:
        continue
    first10 = int(np.argmax(tags == 10)) if (tags == 10).any() else len(tags)
    print(f"wave {w}: {len(ts)} stamps; whole unit {ts[-1]-ts[0]}; prologue (tag: cycles since the stamp before):", [(int(tags[k]), int(ts[k] - ts[k-1])) for k in range(1, first10 + 1) if k < len(ts)])
    i = first10; rows = []
    while i + 2 < len(ts) and len(rows) < 400:
        if tags[i] == 10:
            seq = {}
            k = 1
            while i + k < len(ts) and tags[i + k] != 10:
                seq[int(tags[i + k])] = int(ts[i + k]); k += 1
            nxt = int(ts[i + k]) if i + k < len(ts) else seq.get(13, int(ts[i]))
            rows.append((seq.get(11, ts[i]) - ts[i], seq.get(12, ts[i]) - ts[i], seq.get(13, ts[i]) - ts[i], nxt - ts[i]))
            i += k
        else:
            i += 1
    r = np.array(rows)
    if len(r):
        print(f"   tiles {len(r)}; mean cycles to [A done, B done, barrier passed, next tile] = {r.mean(0).round(0).tolist()}   median {np.median(r,0).tolist()}")
        per = r.reshape(-1, 10, 4) if len(r) % 10 == 0 else None
        if w >= 8 and NWV == 12: per = None
        if per is not None:
            print("   by k tile (mean over c tiles) [A, B, barrier]:", [per[:, k, :3].mean(0).round(0).astype(int).tolist() for k in range(10)])
        print("   first 12 tiles:", r[:12, :3].tolist())
