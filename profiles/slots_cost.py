"""Dev helper: per-frame time of ONE rank's share of a 1080p frame through rm_frame_submit
(layout of N ranks, no transport) with 1..4 frames in flight, on one GPU.  Shows what the
slots' separate streams buy when a rank's share is too small to fill the chip for long.
usage (GPU box): python3 profiles/slots_cost.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per slot (HIP's default of 4 makes slots share)
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G, workloads
pkg = G.load_package()
ctx = pkg.backend.Context(0)
ctx.upload(pkg.Scene.create_default().flatten())
w, h, depth = 1920, 1080, 5
p = pkg.backend.make_params(1.5, float(h), float(w), depth)
f64 = [torch.zeros((h, w, 3), dtype=torch.float64, device="cuda") for _ in range(4)]
for world in [int(x) for x in os.environ.get('WORLDS', '1,2,4,8').split(',')]:
    rows, chunk = ctx.exchange_layout(p, world)
    g8 = [torch.zeros((world * chunk,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    ctx.comm_init(world - 1, world)           # the last rank: owns the bottom (expensive) rows' residue class
    line = []
    for slots in [int(x) for x in os.environ.get('SLOTS', '1,2,3,4').split(',')]:
        n = 600
        for k in range(n + 40):
            if k == 40:
                for b in range(slots):
                    ctx.frame_wait(b)
                t0 = time.perf_counter()
            ctx.frame_submit(p, f64[k % slots].data_ptr(), g8[k % slots].data_ptr(), None, k % slots)
        for b in range(slots):
            ctx.frame_wait(b)
        line.append("%d slot(s) %.1f us" % (slots, (time.perf_counter() - t0) / n * 1e6))
    print("N=%d  rank %d's share per frame: %s" % (world, world - 1, "   ".join(line)))
