"""Diagnostic: occupancy over time from per-wave stamps (RM_EXP_STAMPS build).
usage: python profiles/analyze_stamps.py stamps.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0     # s_memrealtime ticks at 100 MHz -> us
staged = (a[:, 1] - t0).astype(np.float64) / 100.0
end = (a[:, 2] - t0).astype(np.float64) / 100.0
print("waves %d  span %.1f us" % (len(a), end.max()))
life = end - start
print("wave lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (
    life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
st = staged - start
print("staging (launch -> scene in LDS) us: mean %.2f p50 %.2f p90 %.2f" % (st.mean(), *np.percentile(st, [50, 90])))
# occupancy over time
T = np.linspace(0, end.max(), 41)
for lo, hi in zip(T[:-1], T[1:]):
    mid = (lo + hi) / 2
    occ = ((start <= mid) & (end > mid)).sum()
    print("t=%6.1f us  resident waves %5d  (%.2f per SIMD)" % (mid, occ, occ / 1024.0))
hw = (a[:, 3] >> np.uint64(32)).astype(np.uint32)
xcc = (a[:, 3] & np.uint64(0xF)).astype(np.uint32)
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("waves per XCC:", np.bincount(xcc, minlength=8))
