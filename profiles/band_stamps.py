"""Dev helper: wave stamps of ONE rank's share of a 1080p frame in the N-rank layout (cyclic patch
rows), one launch on an otherwise idle GPU: what bounds a single frame when a rank's share is
about one wave per SIMD slot.  Needs the stamps build:
  RM_LIB_PATH=rusty-marcher_amd/lib/variants/stamps/librusty_marcher_amd.so RM_DEBUG_STAMPS=/tmp/s.bin \
      python3 profiles/band_stamps.py 8 && python3 profiles/analyze_stamps.py /tmp/s.bin"""
import sys
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G
pkg = G.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = pkg.backend.Context(0)
ctx.upload(pkg.Scene.create_default().flatten())
w, h, depth = 1920, 1080, 5
P = h // 32
f64 = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
u8 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
p = pkg.backend.make_params(1.5, float(h), float(w), depth, (N - 1, P, N))
p.flags = 4
for _ in range(3):
    ctx.render_device_u8(p, f64.data_ptr(), u8.data_ptr())
    torch.cuda.synchronize()
