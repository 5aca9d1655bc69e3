import sys, numpy as np
f = sys.argv[1]
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 4)
x = np.fromfile(f + ".ext", dtype=np.uint64).reshape(-1, 2)
life = (a[:, 2].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.
ch = (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64) / 100.; sh = (a[:, 3] >> np.uint64(32)).astype(np.int64) / 100.
su = (x[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64) / 100.; sd = (x[:, 1] >> np.uint64(32)).astype(np.int64) / 100.
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(int)
r = (x[:, 0] != np.uint64(0xFFFFFFFF)) & (su > 0)
tot = life[r].sum()
print("tile waves %d of %d; span %.1f us; wave-time %.0f us" % (r.sum(), len(a), (a[:, 2].max() - a[a[:, 0] > 0, 0].min()) / 100., tot))
def row(name, m):
    t = life[m].sum()
    print("  %-22s %6d waves %5.1f %% of wave-time | life %6.1f = start-up %4.1f + closest_hit %5.1f + shadow walks %5.1f + shading proper %5.1f + rest %5.1f us (steps %.1f)" % (
        name, m.sum(), 100 * t / tot, life[m].mean(), su[m].mean(), ch[m].mean(), sh[m].mean(), (sd - sh)[m].mean(), (life - su - ch - sd)[m].mean(), steps[m].mean()))
row("all", r)
for lo, hi in [(0, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 200), (200, 10000)]:
    m = r & (life >= lo) & (life < hi)
    if m.any(): row("life %d-%d us" % (lo, hi), m)
