"""Diagnostic: occupancy over time and ray-step statistics from per-wave stamps
(RM_EXP_STAMPS build: make -C rusty-marcher_amd/csrc variant NAME=stamps DEFS=-DRM_EXP_STAMPS).
Per wave: start time, (steps | rays << 16 | rays handed over << 40) of its tile, end time, hardware id.
usage: RM_LIB_PATH=.../variants/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=stamps.bin python bench.py --steps 1 ...
       python profiles/analyze_stamps.py stamps.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0     # s_memrealtime ticks at 100 MHz -> us
end = (a[:, 2] - t0).astype(np.float64) / 100.0
print("waves %d  span %.1f us" % (len(a), end.max()))
life = end - start
print("wave lifetime us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % (
    life.mean(), *np.percentile(life, [10, 50, 90]), life.max()))
steps = (a[:, 1] & np.uint64(0xFFFF)).astype(np.float64)
rays = ((a[:, 1] >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
taken = (a[:, 1] >> np.uint64(40)).astype(np.float64)
print("ray steps per tile: mean %.2f  p50 %.0f  p90 %.0f  max %.0f" % (steps.mean(), *np.percentile(steps, [50, 90]), steps.max()))
print("rays per tile: mean %.1f (%.2f per pixel)  max %.0f;  lanes with a ray per step: %.1f %%" % (
    rays.mean(), rays.mean() / 64, rays.max(), 100 * rays.sum() / (64 * steps.sum())))
print("rays handed over to idle lanes: %.1f %% of all rays; steps if every lane walked its own tree: unknown here, "
      "lower bound of steps = ceil(rays/64): mean %.2f" % (100 * taken.sum() / rays.sum(), np.ceil(rays / 64).mean()))
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 1000)):
    m = (steps >= lo) & (steps <= hi)
    print("  tiles with %d..%d steps: %5.1f %%  of wave time %5.1f %%" % (lo, hi, 100 * m.mean(), 100 * life[m].sum() / life.sum()))
cs = (a[:, 3] & np.uint64(0xFFFF)).astype(np.float64)
cand = ((a[:, 3] >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64)
wide = ((a[:, 3] >> np.uint64(40)) & np.uint64(0xFFF)).astype(np.float64)
narrow = (a[:, 3] >> np.uint64(52)).astype(np.float64)
if cs.sum() > 0 or wide.sum() > 0:
    print("bundles per tile: %.2f narrow (culled), %.2f wide (plain walk / hierarchy); cull steps per narrow bundle %.2f; "
          "candidates kept per narrow bundle %.2f" % (narrow.mean(), wide.mean(), cs.sum() / max(narrow.sum(), 1), cand.sum() / max(narrow.sum(), 1)))
    one = steps == 1
    print("  tiles with one step: candidates per bundle %.2f" % (cand[one].sum() / max(narrow[one].sum(), 1)))
T = np.linspace(0, end.max(), 21)
for lo, hi in zip(T[:-1], T[1:]):
    mid = (lo + hi) / 2
    occ = ((start <= mid) & (end > mid)).sum()
    print("t=%6.1f us  resident waves %5d  (%.2f per SIMD)" % (mid, occ, occ / 1024.0))
if cs.sum() > 0:
    for nb in (1, 2, 3):
        m = narrow == nb
        if m.any():
            print("  tiles with %d narrow bundle(s): %5.1f %% of tiles, candidates per bundle: mean %.2f  p10 %.0f  p50 %.0f  p90 %.0f" % (
                nb, 100 * m.mean(), (cand[m] / nb).mean(), *np.percentile(cand[m] / nb, [10, 50, 90])))
