"""When do the classes of waves of a launch with a dispatch order start and end?  (stamps build; RM_DEBUG_STAMPS=<file>)
    python3 profiles/order_stamps.py <file> <cls_blocks> <n_sorters> <n_static> <tail_patches> [patches]
classifying workgroups | sorting workgroups | first round | tile waves in order (by eighths) | sky tail"""
import sys
import numpy as np
f, cls, nsort, ns, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
npatch = int(sys.argv[6]) if len(sys.argv) > 6 else 1980
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 4)
t0 = a[a[:, 0] > 0, 0].min()
st = (a[:, 0].astype(np.int64) - int(t0)) / 100.; en = (a[:, 2].astype(np.int64) - int(t0)) / 100.
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(int)
print("waves", len(a), "span %.1f us" % en.max())
print(" classifying: start %.1f..%.1f end p50 %.1f max %.1f  life p50 %.1f max %.1f" % (st[:cls].min(), st[:cls].max(), np.median(en[:cls]), en[:cls].max(), np.median((en - st)[:cls]), (en - st)[:cls].max()))
w = a[:cls, 1].astype(np.int64)
print("   of which: classified after p50 %.1f max %.1f us; slots taken after p50 %.1f max %.1f us" % (np.median(w & 0xFFFFF) / 100., (w & 0xFFFFF).max() / 100., np.median((w >> 20) & 0xFFFFF) / 100., ((w >> 20) & 0xFFFFF).max() / 100.))
s = slice(cls, cls + nsort)
print(" sorting:", " ".join("%.1f-%.1f (keys all there %.1f, order out %.1f, next launch's table %.1f)" % (x, y, x + (int(w) & 0xFFFFF) / 100., x + ((int(w) >> 20) & 0xFFFFF) / 100., x + ((int(w) >> 40) & 0xFFFFF) / 100.)
                           for x, y, w in zip(st[s], en[s], a[s, 1])))
o = cls + nsort
s = slice(o, o + ns)
print(" first round: start %.1f..%.1f end p50 %.1f max %.1f life p50 %.1f" % (st[s].min(), st[s].max(), np.median(en[s]), en[s].max(), np.median((en - st)[s])))
n_dyn = npatch - ns // 16; behind = 16 * (n_dyn - T)
s = slice(o + ns, o + ns + behind); life = (en - st)[s]
print(" tile waves in order %d: start p1 %.1f p50 %.1f p99 %.1f; end max %.1f; life p10 %.1f p50 %.1f p90 %.1f; sky %d life p50 %.2f" % (
    behind, *np.percentile(st[s], [1, 50, 99]), en[s].max(), *np.percentile(life, [10, 50, 90]), (steps[s] == 0).sum(), np.median(life[steps[s] == 0]) if (steps[s] == 0).any() else 0))
for lo in range(0, behind - behind // 8 + 1, behind // 8):
    ss = slice(o + ns + lo, o + ns + lo + behind // 8)
    print("   ids %5d..: start %.1f-%.1f life p50 %.1f steps mean %.2f" % (lo, st[ss].min(), st[ss].max(), np.median((en - st)[ss]), steps[ss].mean()))
s = slice(o + ns + behind, o + ns + behind + T)
if T: print(" tail %d: start %.1f..%.1f life p50 %.1f max %.1f end max %.1f" % (T, st[s].min(), st[s].max(), np.median((en - st)[s]), (en - st)[s].max(), en[s].max()))
