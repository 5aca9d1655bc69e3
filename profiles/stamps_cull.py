import sys, numpy as np
f = sys.argv[1]
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 4)
life = (a[:, 2].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.
c = a[:, 3]
steps = (c & np.uint64(0xFFFF)).astype(np.int64); kept = ((c >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.int64)
wide = ((c >> np.uint64(40)) & np.uint64(0xFFF)).astype(np.int64); narrow = ((c >> np.uint64(52)) & np.uint64(0xFFF)).astype(np.int64)
r = (narrow + wide) > 0
print("waves %d, tracing %d; span %.1f us" % (len(a), r.sum(), (a[:, 2].max() - a[a[:, 0] > 0, 0].min()) / 100.))
print("totals: cull steps %d candidates kept %d wide bundles %d narrow bundles %d" % (steps.sum(), kept.sum(), wide.sum(), narrow.sum()))
tot = life[r].sum()
print("wave-time %.0f us; per bundle %.2f us; per narrow: steps %.2f kept %.2f" % (tot, tot / (wide.sum() + narrow.sum()), steps.sum() / max(narrow.sum(), 1), kept.sum() / max(narrow.sum(), 1)))
for lo, hi in [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 200), (200, 1000)]:
    m = r & (life >= lo) & (life < hi)
    if m.any(): print("  life %3d-%4d us: %6d waves, %5.1f %% of wave-time; per wave: narrow %.1f wide %.1f steps %.1f kept %.1f; us per bundle %.2f" % (lo, hi, m.sum(), 100 * life[m].sum() / tot, narrow[m].mean(), wide[m].mean(), steps[m].mean(), kept[m].mean(), life[m].sum() / (narrow[m].sum() + wide[m].sum())))
