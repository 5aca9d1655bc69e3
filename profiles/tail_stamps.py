"""Which waves does a launch end with?  From per-wave stamps (RM_EXP_STAMPS build, RM_DEBUG_STAMPS=file):
    python3 profiles/tail_stamps.py stamps.bin [n_cls_blocks]
prints, for the waves that end in the last 15 % of the launch, when they started, how many ray steps their tile took and where
in the dispatch order they were (id = blockIdx.x less the classification workgroups at the head of the launch)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
cls = int(sys.argv[2]) if len(sys.argv) > 2 else 0
idx = np.arange(len(a))
ok = a[:, 2] > 0
a, idx = a[ok], idx[ok]
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0
end = (a[:, 2] - t0) / 100.0
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(int)
span = end.max()
print("waves %d span %.1f us; classification workgroups at the head: %d" % (len(a), span, cls))
late = end > 0.85 * span
print("waves ending in the last 15 %%: %d; their ray steps: %s" % (late.sum(), np.bincount(steps[late])))
print("their starts (us): p10 %.1f p50 %.1f p90 %.1f;  their dispatch ids: p10 %d p50 %d p90 %d of %d" % (
    *np.percentile(start[late], [10, 50, 90]), *np.percentile(idx[late] - cls, [10, 50, 90]), len(a) - cls))
for lo in range(0, 100, 10):
    m = (end > span * lo / 100.) & (end <= span * (lo + 10) / 100.)
    real = m & (steps > 0)
    print("ends in %3d-%3d %%: %6d waves, %6d with rays, mean steps of those %.2f, mean life %.1f us" % (
        lo, lo + 10, m.sum(), real.sum(), steps[real].mean() if real.any() else 0., (end - start)[real].mean() if real.any() else 0.))
# when is each tenth of the dispatch order started?
order = np.argsort(idx)
for lo in range(0, 100, 10):
    sel = order[int(len(a) * lo / 100.):int(len(a) * (lo + 10) / 100.)]
    print("dispatch ids %3d-%3d %%: started %.1f..%.1f us, steps mean %.2f, life mean %.1f" % (
        lo, lo + 10, start[sel].min(), start[sel].max(), steps[sel].mean(), (end - start)[sel].mean()))
# where are the deep tiles?  (bottom-up dispatch: tile = last - id; 16 tiles per 32x32 patch, patches row-major from the top)
if len(sys.argv) > 3:
    n_width = int(sys.argv[3])
    n_tiles = len(a) - cls
    tid = (n_tiles - 1) - (idx - cls)
    real = (idx >= cls)
    prow = (tid >> 4) // n_width
    print("patch row (from the top): tiles with rays / with >= 3 steps / with >= 5 steps / mean life us / first start us")
    for r in range(int(prow[real].max()) + 1):
        m = real & (prow == r)
        if not m.any():
            continue
        print("  row %2d: %4d %4d %4d  %5.1f  %5.1f" % (r, (steps[m] > 0).sum(), (steps[m] >= 3).sum(), (steps[m] >= 5).sum(),
                                                       (end - start)[m & (steps > 0)].mean() if (m & (steps > 0)).any() else 0., start[m].min()))
