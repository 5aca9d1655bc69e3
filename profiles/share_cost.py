import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G
pkg = G.load_package()
ctx = pkg.backend.Context(0)
ctx.upload(pkg.Scene.create_default().flatten())
w, h, depth = 1920, 1080, 5
p = pkg.backend.make_params(1.5, float(h), float(w), depth)
f64 = [torch.zeros((h, w, 3), dtype=torch.float64, device="cuda") for _ in range(4)]
for world in (1, 8):
    rows, chunk = ctx.exchange_layout(p, world)
    g8 = [torch.zeros((world * chunk,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    ctx.comm_init(world - 1, world)
    line = []
    for slots in (1, 4):
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
    print("N=%d share per frame: %s" % (world, "   ".join(line)))
