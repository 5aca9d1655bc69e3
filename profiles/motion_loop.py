"""Per-frame time of a render loop whose camera moves a little every frame (one stream, device-resident f64 + display frames):
    python3 profiles/motion_loop.py C3        (RM_SKY_TAIL_MOTION=0: no guessed sky tail; STEPS=0,1,4 camera step sizes; NFR frames)
-> us per frame by HIP events and in how many launches the sky tail was armed."""
import os, sys, time
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G, workloads
pkg = G.load_package()
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = workloads.CONFIGS[cfgname]
ctx = pkg.backend.Context(0)
scene = workloads.product_scene(pkg, cfg["scene"])
ctx.upload(scene.flatten())
w, h = cfg["width"], cfg["height"]
p = pkg.backend.make_params(workloads.FOV, float(h), float(w), cfg["max_depth"])
f64 = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
u8 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
L = pkg.lib()
def frame(k, step):
    cam = pkg.Vec3f(step * (k % 40 - 20) * 0.05 + 0., step * ((k * 7) % 30 - 15) * 0.02 + 0., 0.)
    ctx.set_camera(cam)
    ctx.render_device_u8(p, f64.data_ptr(), u8.data_ptr(), stream.cuda_stream)
for step in [float(x) for x in os.environ.get('STEPS', '0,1,4').split(',')]:
    for k in range(60): frame(k, step)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = int(os.environ.get('NFR', '400')); tails = 0
    e0.record(stream)
    for k in range(n):
        frame(60 + k, step); tails += ctx.launch_stats()[1] > 0
    e1.record(stream); torch.cuda.synchronize()
    print("%s camera step %.0f: %.1f us per frame, tail in %d of %d launches" % (cfgname, step, e0.elapsed_time(e1) / n * 1e3, tails, n))
