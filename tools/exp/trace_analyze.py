"""Analyse a DBFR_CONV_TRACE dump: per-tile timeline of k_conv waves (developer tool)."""
import sys
import numpy as np
TB, TT = 1024, 48
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(TB, 4, 2 + 3 * TT + 8)
ph = a[:, :, 2 + 3 * TT:2 + 3 * TT + 6].astype(np.int64)
rt = a[:, :, 2 + 3 * TT + 6:].astype(np.int64)
a = a[:, :, :2 + 3 * TT]
hw, xcc = a[:, :, 0].astype(np.int64), a[:, :, 1].astype(np.int64) & 15
t = a[:, :, 2:].astype(np.int64).reshape(TB, 4, TT, 3)       # start, mfma done, epilogue done
ok = t[:, :, :, 2].min(axis=2) > 0
mf = (t[..., 1] - t[..., 0])[ok]
ep = (t[..., 2] - t[..., 1])[ok]
per = (t[:, :, 1:, 0] - t[:, :, :-1, 0])[ok]
print("waves traced", ok.sum(), " MFMA phase cycles: median %d p10 %d p90 %d" % (np.median(mf), np.percentile(mf, 10), np.percentile(mf, 90)))
print("epilogue cycles: median %d p10 %d p90 %d" % (np.median(ep), np.percentile(ep, 10), np.percentile(ep, 90)))
print("tile period: median %d  (ideal 2 waves/SIMD x 108 MFMA x 32 = 6912 if clock unit = core cycles)" % np.median(per))
# pairs on the same SIMD
key = (xcc * 1000000 + ((hw >> 8) & 0xff) * 100 + ((hw >> 13) & 7) * 10000 + ((hw >> 4) & 3))
first = {}
pairs = []
for b in range(TB):
    for w in range(4):
        if not ok[b, w]:
            continue
        k = (int(key[b, w]), )
        # co-resident = overlapping in time
        for (b2, w2) in first.get(k, []):
            if abs(int(t[b, w, 0, 0]) - int(t[b2, w2, 0, 0])) < 200000:
                pairs.append((b2, w2, b, w))
        first.setdefault(k, []).append((b, w))
print("co-resident pairs found", len(pairs))
offs, over = [], []
for (b1, w1, b2, w2) in pairs[:4000]:
    p = np.median(t[b1, w1, 1:, 0] - t[b1, w1, :-1, 0])
    # phase offset of wave 2's tile starts relative to wave 1's, in fractions of the period
    d = (t[b2, w2, 10:30, 0][:, None] - t[b1, w1, 5:40, 0][None, :])
    d = d[(d >= 0)].reshape(-1)
    if len(d) == 0:
        continue
    offs.append(float(np.min(d[d >= 0]) / p) if p > 0 else 0)
    # epilogue overlap: fraction of wave-1 epilogue intervals that intersect a wave-2 epilogue interval
    e1 = t[b1, w1, 5:40, 1:3]; e2 = t[b2, w2, :, 1:3]
    ov = 0
    for s, e in e1:
        ov += np.any((e2[:, 0] < e) & (e2[:, 1] > s))
    over.append(ov / len(e1))
offs = np.array(offs); over = np.array(over)
print("phase offset histogram (fraction of period):", np.histogram(offs % 1.0, bins=10, range=(0, 1))[0])
print("mean fraction of epilogues overlapping the partner's epilogue: %.3f" % over.mean())
if not pairs:
    pairs = [(0, 0, 1, 0)]
b, w = pairs[0][0], pairs[0][1]
b2, w2 = pairs[0][2], pairs[0][3]
base = min(t[b, w, 0, 0], t[b2, w2, 0, 0])
print("example pair timelines (start, mfma_done, epi_done) relative cycles:")
for i in range(8, 14):
    print("  A", (t[b, w, i] - base).tolist(), "  B", (t[b2, w2, i] - base).tolist())

# ---- phases per wave: entry, gather done, B operand ready (start of D), D done, after the D barrier, end
okp = ph.min(axis=2) > 0
d = np.diff(ph, axis=2)[okp]
names = ["A gather", "B hidden layer + C", "D tiles", "wait at barrier", "E store"]
tot = (ph[..., 5] - ph[..., 0])[okp]
print("workgroup life (cycles): median %d" % np.median(tot))
for i, nm in enumerate(names):
    print("  %-20s median %7d  mean %7d  (%.1f%% of life)" % (nm, np.median(d[:, i]), d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))

# wall time of phase D from s_memrealtime (100 MHz) next to the shader-clock ticks: effective clock and ns per tile
okr = (rt.min(axis=2) > 0) & okp
dns = (rt[..., 1] - rt[..., 0])[okr] * 10.0
dtk = (ph[..., 3] - ph[..., 2])[okr]
print("phase D: median %.0f ns = %d ticks => shader clock %.3f GHz during phase D" % (np.median(dns), np.median(dtk), np.median(dtk / dns)))
# ---- co-resident workgroups: offset between their starts and how much of one's non-D time the other spends in D
wg = {}
for b in range(TB):
    if okp[b, 0]:
        wg[b] = (int(xcc[b, 0]), int((hw[b, 0] >> 8) & 0xff), int((hw[b, 0] >> 13) & 7), ph[b, 0])
by_cu = {}
for b, (x, cu, se, p) in wg.items():
    by_cu.setdefault((x, cu, se), []).append((int(p[0]), b))
offs, cover = [], []
for key, lst in by_cu.items():
    lst.sort()
    for i in range(len(lst)):
        for j in range(i + 1, len(lst)):
            a_, b_ = ph[lst[i][1], 0], ph[lst[j][1], 0]
            if b_[0] >= a_[5]:
                continue                      # not co-resident
            offs.append(int(b_[0] - a_[0]))
            # B's non-D intervals: [0,2] and [3,5]; A's D interval: [2,3]
            nd = (b_[2] - b_[0]) + (b_[5] - b_[3])
            ov = max(0, min(b_[2], a_[3]) - max(b_[0], a_[2])) + max(0, min(b_[5], a_[3]) - max(b_[3], a_[2]))
            cover.append(ov / max(nd, 1))
offs = np.array(offs)
print("co-resident workgroup pairs: %d; start offset (ticks) p10 %d median %d p90 %d; fraction of the later one's non-D time inside the earlier one's D phase: mean %.2f"
      % (len(offs), np.percentile(offs, 10), np.median(offs), np.percentile(offs, 90), np.mean(cover)))
