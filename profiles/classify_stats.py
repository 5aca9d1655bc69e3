"""How many tiles the classification launch (rm_classify.hip) lists per workload, and -- with
RM_DEBUG_CLASSIFY=1 in the environment -- which primitives stay in their masks (stderr).
    RM_DEBUG_CLASSIFY=1 python3 profiles/classify_stats.py   (GPU box, from the repo root)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G   # noqa: E402
import workloads              # noqa: E402

pkg = G.load_package()
ctx = pkg.backend.Context(0)
for name, w, h, d in (("demo", 1920, 1080, 5), ("demo", 7680, 4320, 8), ("cornell", 1920, 1080, 5), ("synthetic256", 4096, 4096, 10)):
    ctx.upload(workloads.product_scene(pkg, name).flatten())
    p = pkg.backend.make_params(workloads.FOV, float(h), float(w), d)
    ctx.render(p, None)
    tiles, listed = ctx.tile_stats()
    print("%-13s %5dx%-5d tiles %7d listed %7d (%.1f %%)" % (name, w, h, tiles, listed, 100. * listed / max(tiles, 1)), flush=True)
