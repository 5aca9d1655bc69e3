"""Dev helper: kernel time of one rank's share of a 1080p frame for N = 1, 2, 4, 8 ranks
(cyclic rows vs contiguous bands), measured on one GPU with HIP events."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch, numpy as np
import __graft_entry__ as G, workloads
pkg = G.load_package()
ctx = pkg.backend.Context(0)
ctx.upload(pkg.Scene.create_default().flatten())
w, h, depth = 1920, 1080, 5
P = h // 32
f64 = torch.zeros((h + 64, w, 3), dtype=torch.float64, device="cuda")
u8 = torch.zeros((h + 64, w, 3), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
def t(band, flags=0, n=200):
    p = pkg.backend.make_params(1.5, float(h), float(w), depth, band); p.flags = flags
    for _ in range(20): ctx.render_device_u8(p, f64.data_ptr(), u8.data_ptr(), s.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): ctx.render_device_u8(p, f64.data_ptr(), u8.data_ptr(), s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("full frame: %.1f us" % t((0, P)))
for N in (2, 4, 8):
    cyc = [t((r, P, N), 4) for r in range(N)]
    c, bands = workloads.equal_bands(P, N)
    con = [t(b) if b[1] > b[0] else 0. for b in bands]
    print("N=%d cyclic   per-rank us: %s  max %.1f" % (N, " ".join("%.1f" % x for x in cyc), max(cyc)))
    print("N=%d contiguous per-rank us: %s  max %.1f" % (N, " ".join("%.1f" % x for x in con), max(con)))
