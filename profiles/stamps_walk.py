import sys, numpy as np
f = sys.argv[1]
a = np.fromfile(f, dtype=np.uint64).reshape(-1, 4)
x = np.fromfile(f + ".ext", dtype=np.uint64).reshape(-1, 2)
life = (a[:, 2].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.
walks = (a[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64); nodes = (a[:, 3] >> np.uint64(32)).astype(np.int64)
prims = (x[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64); hits = (x[:, 1] >> np.uint64(32)).astype(np.int64)
r = walks > 0
print("waves %d, with walks %d; span %.1f us" % (len(a), r.sum(), (a[:, 2].max() - a[a[:, 0] > 0, 0].min()) / 100.))
print("totals: walks %d nodes %d leaf prims %d lane-hits %d" % (walks.sum(), nodes.sum(), prims.sum(), hits.sum()))
print("per walk: nodes %.1f, leaf prims %.1f; lanes meeting a child box per node %.1f of 128" % (nodes.sum() / walks.sum(), prims.sum() / walks.sum(), hits.sum() / nodes.sum()))
print("per wave with walks: walks p50 %d p90 %d max %d; life p50 %.1f p90 %.1f max %.1f us" % (*np.percentile(walks[r], [50, 90]), walks.max(), *np.percentile(life[r], [50, 90]), life.max()))
tot = life[r].sum()
print("wave-time %.0f us total; per walk %.3f us; per node %.3f us (all of a wave's time put on its nodes)" % (tot, tot / walks.sum(), tot / nodes.sum()))
for lo, hi in [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1000)]:
    m = r & (life >= lo) & (life < hi)
    if m.any(): print("  life %3d-%3d us: %6d waves, %5.1f %% of wave-time, walks/wave %.0f nodes/walk %.1f prims/walk %.1f lanes/node %.1f" % (lo, hi, m.sum(), 100 * life[m].sum() / tot, walks[m].mean(), nodes[m].sum() / walks[m].sum(), prims[m].sum() / walks[m].sum(), hits[m].sum() / max(nodes[m].sum(), 1)))
