"""Timeline of k_conv2r's workgroup 0 (developer):
    DBFR_CONV2_TRACE=/tmp/t.bin DBFR_CONV2R_ABL=128 DBFR_CONV2=1 DBFR_GEMM=split python tools/conv_bench.py --layer 3 --fam 2 --edges 650000 --reps 1
    python tools/exp/ring_trace.py /tmp/t.bin
Stamps are (tag << 56 | shader clock) per wave: 0x1s arrival at k-step s, 0x2s own LDS operations done, 0x3s barrier passed (s = 0, 2, 4),
0x40 tile done, 0x50 / 0x51 run-end contraction, 0x01 unit prologue done.  Prints mean shader cycles per segment over the recorded tiles."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8, -1)
for w in range(8):
    v = raw[w][raw[w] != 0]
    tag, clk = (v >> np.uint64(56)).astype(int), (v & np.uint64((1 << 56) - 1)).astype(np.int64)
    seg = {}
    tiles = []
    t0 = None
    for i in range(1, len(tag)):
        a, b, d = tag[i - 1], tag[i], int(clk[i] - clk[i - 1])
        seg.setdefault((a, b), []).append(d)
        if b == 0x10:
            t0 = clk[i]
        if b == 0x40 and t0 is not None:
            tiles.append(int(clk[i] - t0))
    if w in (0, 4):
        print(f"wave {w}: {len(tiles)} tiles, mean tile {np.mean(tiles):.0f} cycles (median {np.median(tiles):.0f})")
        for k in sorted(seg, key=lambda k: -np.sum(seg[k]))[:16]:
            print(f"   {k[0]:#04x} -> {k[1]:#04x}: n={len(seg[k]):5d} mean {np.mean(seg[k]):7.0f}  median {np.median(seg[k]):7.0f}  total share {np.sum(seg[k]) / max(1, clk[-1] - clk[0]):.3f}")
