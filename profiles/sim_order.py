"""List scheduling of one launch's measured wave lifetimes (a stamps file: RM_EXP_STAMPS build, RM_DEBUG_STAMPS=<file>) over
4,096 wave slots, in the order the launch dispatched them and in the orders it could have: every tile sorted on its own,
the patches by their longest tile, the patches by the sum of their tiles.
    python profiles/sim_order.py gpurun_out/st_C2_0.bin      (a launch of 495 classifying + 1 sorting workgroup in front)
A wave's lifetime depends on what it shares its SIMD with, so this is an estimate -- it was within 3 us of the measured
launch for the order the launch had, and what it said about the others held: 73.5 us by the sum, 66.8 by the longest tile."""
import sys, heapq
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
cls, srt = 495, 1
t0 = a[a[:, 0] > 0, 0].min()
start = (a[:, 0].astype(np.int64) - int(t0)) / 100.0
end = (a[:, 2].astype(np.int64) - int(t0)) / 100.0
life = end - start
print("actual span %.1f" % end.max())
tiles = life[cls + srt:]
head = life[:cls + srt]
def sim(order_lives, slots=4096, dispatch_ns=0.0):
    # list scheduling: waves dispatched in order to the earliest free slot
    h = [0.0] * slots
    heapq.heapify(h)
    t_disp = 0.0
    endmax = 0.0
    for L in order_lives:
        t = heapq.heappop(h)
        t = max(t, t_disp)
        t_disp = t + dispatch_ns * 1e-3
        e = t + L
        endmax = max(endmax, e)
        heapq.heappush(h, e)
    return endmax
cur = np.concatenate([head, tiles])
print("sim current order        %.1f" % sim(cur))
print("sim current, 0.25ns/wave %.1f" % sim(cur, dispatch_ns=0.25))
lpt = np.concatenate([head, np.sort(tiles)[::-1]])
print("sim tile LPT             %.1f" % sim(lpt))
print("sim tile LPT 0.25ns      %.1f" % sim(lpt, dispatch_ns=0.25))
# patch-level LPT by max tile in patch (tiles are in dispatch order: 16 per patch consecutive)
p = tiles[: len(tiles) // 16 * 16].reshape(-1, 16)
o = np.argsort(-p.max(axis=1))
print("sim patch by max tile    %.1f" % sim(np.concatenate([head, p[o].ravel()])))
o = np.argsort(-p.sum(axis=1))
print("sim patch by sum (ideal) %.1f" % sim(np.concatenate([head, p[o].ravel()])))
print("sum/4096 = %.1f" % (cur.sum() / 4096))
