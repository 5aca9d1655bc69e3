"""Diagnostic: per-launch kernel time of the first launches after a device-wide synchronize
(the contract's fence), 1080p demo frame.  One HIP event between consecutive launches.
usage (GPU box): python profiles/after_fence.py [idle_ms ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import __graft_entry__ as G
import workloads

pkg = G.load_package()
L = pkg.lib()
cfg = workloads.CONFIGS["C2"]
w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
ctx = pkg.backend.Context(0)
ctx.upload(workloads.product_scene(pkg, "demo").flatten())
p = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)
frame = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
frame8 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.Stream(device="cuda:0")
torch.cuda.set_stream(stream)
sp = C.c_void_p(stream.cuda_stream)


def launch():
    st = L.rm_render_device_u8(ctx.ptr, C.byref(p), C.c_void_p(frame.data_ptr()), C.c_void_p(frame8.data_ptr()), sp)
    assert st == 0


for idle_ms in [float(a) for a in sys.argv[1:]] or [0., 1., 20.]:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        launch()
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    n = 24
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(n):
        launch()
        ev[k + 1].record(stream)
    t_submit = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
    print("idle %5.1f ms after the fence: submit %.0f us, wall %.0f us (%.1f us per launch); per launch us: %s" % (
        idle_ms, t_submit * 1e6, wall * 1e6, wall * 1e6 / n, " ".join("%.0f" % x for x in per)))

# the trend over a long burst: one event per 10 launches
for rep in range(2):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        launch()
    torch.cuda.synchronize()
    torch.cuda.synchronize()
    n, g = 400, 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n // g + 1)]
    ev[0].record(stream)
    for k in range(n):
        launch()
        if (k + 1) % g == 0:
            ev[(k + 1) // g].record(stream)
    torch.cuda.synchronize()
    per = [ev[k].elapsed_time(ev[k + 1]) * 1e3 / g for k in range(n // g)]
    print("400 launches after a fence, us per launch in groups of 10: %s" % " ".join("%.1f" % x for x in per))
