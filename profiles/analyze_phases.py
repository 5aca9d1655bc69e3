"""Diagnostic: share of wave time per phase (RM_EXP_PHASES build; shader-clock cycles)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4).astype(np.float64)
a = a[a.sum(axis=1) > 0]
tot = a.sum()
names = ["closest_hit", "shade_direct (incl. shadow rays, surface)", "children + loop control", "setup + store"]
print("waves %d, mean cycles per wave %.0f" % (len(a), a.sum(axis=1).mean()))
for i, n in enumerate(names):
    print("%-45s %5.1f %%   mean %.0f cycles" % (n, 100 * a[:, i].sum() / tot, a[:, i].mean()))
