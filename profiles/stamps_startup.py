import sys, numpy as np
f = sys.argv[1]
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 4)
x = np.fromfile(f + ".ext", dtype=np.uint64).reshape(-1, 2)
life = (a[:, 2].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.
c0 = (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); c1 = (a[:, 3] >> np.uint64(32)).astype(np.int64)
c2 = (x[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t_tile = (c0 & 0xFFFF) / 100.; t_mask = (c0 >> 16) / 100.; t_staged = (c1 & 0xFFFF) / 100.; t_loop = (c1 >> 16) / 100.; t_end = c2 / 100.
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(int)
r = (x[:, 0] != np.uint64(0xFFFFFFFF)) & (t_loop > 0) & (t_end > 0)
st = (a[:, 0].astype(np.int64) - int(a[a[:, 0] > 0, 0].min())) / 100.
print("tile waves %d of %d; span %.1f us" % (r.sum(), len(a), (a[:, 2].max() - a[a[:, 0] > 0, 0].min()) / 100.))
def row(name, m):
    print("  %-26s %6d waves | tile known %.2f, +%.2f its word, +%.2f the scene copy, +%.2f to the first ray step (= %.2f us) | ray steps %.1f us (%.1f steps) | stores and leaving %.2f us" % (
        name, m.sum(), t_tile[m].mean(), (t_mask - t_tile)[m].mean(), (t_staged - t_mask)[m].mean(), (t_loop - t_staged)[m].mean(), t_loop[m].mean(), (t_end - t_loop)[m].mean(), steps[m].mean(), (life - t_end)[m].mean()))
row("all", r)
for lo, hi in [(0, 3), (3, 10), (10, 30), (30, 60), (60, 2000)]:
    m = r & (st >= lo) & (st < hi)
    if m.any(): row("started %d-%d us" % (lo, hi), m)
