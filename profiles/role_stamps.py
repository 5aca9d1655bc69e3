"""Wave stamps of one launch (RM_EXP_STAMPS build, RM_DEBUG_STAMPS=<file>; the library's stderr of that run names the
launch's geometry) by the role of the workgroup: classifying, sorting, a patch of the sky tail, a tile -- when they
started, how long they lived, what they add up to -- and the waves resident over the launch.
    python profiles/role_stamps.py st.bin st.err"""
import sys, re
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
info = open(sys.argv[2]).read()
m = re.findall(r"stamps: grid (\d+) cls_blocks (\d+) sort_block (\d+) n_tiles (\d+) tail_patches (\d+) tail_q (\d+)", info)[-1]
grid, cls, srt, n_tiles, T, q = map(int, m)
print("grid %d cls %d sort %d n_tiles %d tail %d q %d" % (grid, cls, srt, n_tiles, T, q))
a = a[:grid]
t0 = a[a[:, 0] > 0, 0].min()
start = (a[:, 0].astype(np.int64) - int(t0)) / 100.0
end = (a[:, 2].astype(np.int64) - int(t0)) / 100.0
ids = np.arange(grid)
kind = np.full(grid, "tile ", dtype=object)
kind[:cls] = "cls  "
if srt: kind[cls] = "sort "
base = cls + srt
head_ids = n_tiles - 16 * T
fr = min(4096, head_ids)
if T:
    mm = ids - base - fr
    ok = mm >= 0
    mmc = np.where(ok, mm, 0).astype(np.uint64)
    if q:
        before = (mmc * np.uint64(q)) >> np.uint64(32)
        after = ((mmc + np.uint64(1)) * np.uint64(q)) >> np.uint64(32)
        mine = ok & (after > before)
    else:
        mine = ok & (mm < T)
    kind[mine] = "patch"
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(int)
for k in ["cls  ", "sort ", "patch", "tile "]:
    sel = kind == k
    if not sel.any(): continue
    life = (end - start)[sel]
    print("%s n %6d  start %.1f..%.1f us  end max %.1f  life mean %.2f p50 %.2f p90 %.2f max %.2f  sum %.0f" % (k, sel.sum(), start[sel].min(), start[sel].max(), end[sel].max(), life.mean(), *np.percentile(life, [50, 90]), life.max(), life.sum()))
sel = (kind == "tile ")
for lo, hi, name in [(0, 0, "tile 0 steps (sky / exit)"), (1, 1, "tile 1 step"), (2, 100, "tile 2+ steps")]:
    s2 = sel & (steps >= lo) & (steps <= hi)
    if s2.any():
        life = (end - start)[s2]
        print("  %-26s n %6d life mean %.2f p50 %.2f p90 %.2f  start %.1f..%.1f sum %.0f" % (name, s2.sum(), life.mean(), *np.percentile(life, [50, 90]), start[s2].min(), start[s2].max(), life.sum()))
print("span %.1f us" % end.max())
T_ = np.linspace(0, end.max(), 21)
print("resident/1024: " + " ".join("%.2f" % (((start <= (lo + hi) / 2) & (end > (lo + hi) / 2)).sum() / 1024.0) for lo, hi in zip(T_[:-1], T_[1:])))
