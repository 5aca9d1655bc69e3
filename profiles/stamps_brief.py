import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0
end = (a[:, 2] - t0).astype(np.float64) / 100.0
life = end - start
print("waves %d span %.1f us; lifetime mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f; sum of lifetimes %.0f us" % (len(a), end.max(), life.mean(), *np.percentile(life, [10, 50, 90]), life.max(), life.sum()))
T = np.linspace(0, end.max(), 21)
print(" ".join("%.2f" % (((start <= (lo+hi)/2) & (end > (lo+hi)/2)).sum() / 1024.0) for lo, hi in zip(T[:-1], T[1:])))
