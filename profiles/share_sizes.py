"""Per-frame time of one rank's share of the 1080p frame at N = 8 -- 33 patch rows: rank 0 owns five (4,800 tiles), the others four
(3,840) -- through rm_frame_submit (layout of 8 ranks, no transport), one frame at a time and four in flight, one GPU:
    RM_CLASSIFY_MIN_TILES=<n> python3 profiles/share_sizes.py        (what a launch must have to be classified and ordered)
VERDICT r3 item 5: both share sizes should take the same path."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G, workloads
pkg = G.load_package()
w, h, depth, world = 1920, 1080, 5, 8
for rank in (0, 7):
    ctx = pkg.backend.Context(0)
    ctx.upload(pkg.Scene.create_default().flatten())
    p = pkg.backend.make_params(1.5, float(h), float(w), depth)
    f64 = [torch.zeros((h, w, 3), dtype=torch.float64, device="cuda") for _ in range(4)]
    rows, chunk = ctx.exchange_layout(p, world)
    g8 = [torch.zeros((world * chunk,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    ctx.comm_init(rank, world)
    line = []
    for slots in (1, 4):
        n = 800
        for k in range(n + 80):
            if k == 80:
                for b in range(slots):
                    ctx.frame_wait(b)
                t0 = time.perf_counter()
            ctx.frame_submit(p, f64[k % slots].data_ptr(), g8[k % slots].data_ptr(), None, k % slots)
        for b in range(slots):
            ctx.frame_wait(b)
        line.append("%d slot(s) %.1f us" % (slots, (time.perf_counter() - t0) / n * 1e6))
    n_rows = len(range(rank, h // 32, world))
    print("RM_CLASSIFY_MIN_TILES=%s  rank %d of 8: %d patch rows, %d tiles: %s" % (os.environ.get("RM_CLASSIFY_MIN_TILES", "(4608)"), rank, n_rows, n_rows * (w // 32) * 16, "   ".join(line)), flush=True)
    ctx.close()
